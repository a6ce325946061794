#!/bin/bash
export TMPDIR=/tmp
python tools/dev/bisect_emu.py hoisted
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['kernel_table']
print(d['value'], d['compressed_bytes'], d['roundtrip_ok'], 'sum', t['sum_ms_per_block_without_symbol_ranking'], [(r['kernel'][:14], r['avg_launch_us']) for r in t['rows'][1:5]])"
timeout 120 python tools/dev/members_scale.py 8 8 2>&1 | tail -2
