#!/bin/bash
# dev: the whole GPU tier, then the bench line with the per-block schedule shown
export TMPDIR=/tmp
timeout 1500 python -u -m pytest tests -m gpu -q --timeout=500 -x > gpurun_out/r06r_pytest_gpu.log 2>&1
tail -6 gpurun_out/r06r_pytest_gpu.log
ORZ_FAST_SHOWSCHED=1 timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r06r_bench.json 2> gpurun_out/r06r_bench.err
grep "^block" gpurun_out/r06r_bench.err | head -14
python - <<'P'
import json
d=json.load(open('gpurun_out/r06r_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','size_delta_pct','roundtrip_ok','compressed_bytes','host_syncs_per_block')}, d['roofline']['avg_launch_us'], d['members']['value'], d['members_l2_text']['value'], d['members_l2_zeros']['value'], d['kernel_table']['sum_ms_per_block_without_symbol_ranking'], d['config']['encoder'])
P
