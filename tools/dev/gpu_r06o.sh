#!/bin/bash
# dev: bench line only, with the per-block schedule shown
export TMPDIR=/tmp
ORZ_FAST_SHOWSCHED=1 timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r06o_bench.json 2> gpurun_out/r06o_bench.err
grep "^block" gpurun_out/r06o_bench.err | head -8
python - <<'P'
import json
d=json.load(open('gpurun_out/r06o_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','size_delta_pct','roundtrip_ok','compressed_bytes')}, d['roofline']['avg_launch_us'], d['members']['value'], d['members_l2_text']['value'], d['members_l2_zeros']['value'], d['kernel_table']['sum_ms_per_block_without_symbol_ranking'])
P
