#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -u -m pytest tests/test_gpu_fast.py -m gpu -x -q --timeout=250 -k "host_waits" 2>&1 | tail -6
