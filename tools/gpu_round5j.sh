#!/bin/bash
export TMPDIR=/tmp
echo "== default queues"; timeout 200 python tools/dev/members_scale.py 8 12 16 8 2>&1 | tail -4
echo "== GPU_MAX_HW_QUEUES=32"; GPU_MAX_HW_QUEUES=32 timeout 200 python tools/dev/members_scale.py 8 12 16 2>&1 | tail -3
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 200 python tools/dev/members_scale.py 8 16 2>&1 | tail -2
echo "== no graph"; ORZ_GRAPH=0 timeout 100 python tools/dev/members_scale.py 8 2>&1 | tail -1
