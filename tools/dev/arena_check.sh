#!/bin/bash
# dev: one device allocation per encoder (HipBackend arena) against a hipMalloc per buffer: three sets of eight encoders in one
# process (the second and third sets allocate after the first was freed), the bench line, a few tests
for a in 0 5632; do
  export ORZ_ARENA_MB=$a
  python tools/dev/members_scale.py 8 8 8 4 1 2>/dev/null | python3 -c "
import json,sys
print('ORZ_ARENA_MB=$a encoders 8 8 8 4 1 ->', [json.loads(l)['MBps'] for l in sys.stdin])"
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('ORZ_ARENA_MB=$a bench', d['value'], d['ms_per_step'], d['roundtrip_ok'])"
done
unset ORZ_ARENA_MB
timeout 200 python -m pytest tests/test_gpu_fast.py tests/test_gpu_parity.py -m gpu -x -q --timeout=200 2>&1 | tail -2
