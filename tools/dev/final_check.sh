#!/bin/bash
export TMPDIR=/tmp
for nr in 16 8 12; do
ORZ_FAST_NEAR=$nr timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['kernel_table']
print('near=$nr', d['value'], d['compressed_bytes'], d['roundtrip_ok'], 'sum', t['sum_ms_per_block_without_symbol_ranking'], [(r['kernel'][:14], r['avg_launch_us']) for r in t['rows'][1:3]])"
done
