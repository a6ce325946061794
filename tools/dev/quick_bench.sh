#!/bin/bash
# dev: one short bench run (2 steps) -> value, ranking launch time, round trip; plus the symbol-ranking guard test
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/tmp/qb.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'rank_launch_us': d['roofline']['avg_launch_us'], 'roundtrip_ok': d['roundtrip_ok'], 'bytes': d['compressed_bytes']}))"
grep -i "repeated\|error" /tmp/qb.err | head -3
