#!/bin/bash
# dev: FastEval built for 5 (default) / 6 / 7 waves per SIMD (backend_hip.h ORZ_EVAL_WAVES): bench line + eight encoders
for v in default w6 w7; do
  if [ $v = default ]; then unset ORZ_LIB_PATH; else export ORZ_LIB_PATH=$PWD/build/variants/liborz_$v.so; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());fe=[r for r in d['roofline_others'] if 'FastEval' in r['kernel']][0]
print('$v', 'bench', d['value'], 'MB/s; FastEval avg', fe['avg_launch_us'], 'us; parse s/step', d['stage_seconds_per_step']['parse'], 'size', d['compressed_bytes'], d['roundtrip_ok'])"
  python tools/dev/members_scale.py 8 8 2>/dev/null | python3 -c "
import json,sys
print('$v', 'eight encoders MB/s:', [json.loads(l)['MBps'] for l in sys.stdin])"
done
