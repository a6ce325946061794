#!/bin/bash
# dev: the lanes symbol-ranking kernel in the product library -- the parity tiers that exercise it, then the bench line
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast.py tests/test_gpu_verify.py -m gpu -x -q > gpurun_out/r06n_pytest.log 2>&1
tail -5 gpurun_out/r06n_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r06n_bench.json 2> gpurun_out/r06n_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r06n_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','size_delta_pct','roundtrip_ok')}, d['roofline']['avg_launch_us'], d['members']['value'], d['members_l2_text']['value'], d['members_l2_zeros']['value'], d['kernel_table']['sum_ms_per_block_without_symbol_ranking'])
P
