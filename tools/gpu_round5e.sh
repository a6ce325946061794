#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for part in none x8 x8m c8 c8m x4m x2m; do
  if [ $part = none ]; then unset ORZ_CU_PART; else export ORZ_CU_PART=$part; fi
  echo "== ORZ_CU_PART=$part"; timeout 120 python tools/dev/members_scale.py 8 8 2>&1 | tail -2
done
unset ORZ_CU_PART
for n1 in 16 8 4 0; do
  echo "== near1=$n1"; ORZ_FAST_NEAR1=$n1 timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['kernel_table']
print(d['value'], d['compressed_bytes'], d['roundtrip_ok'], t['sum_ms_per_block_without_symbol_ranking'], [(r['kernel'][:12], r['avg_launch_us']) for r in t['rows'][1:4]])"
done
