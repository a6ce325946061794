#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -u -m pytest tests/test_gpu_soak.py tests/test_gpu_fast.py -m gpu -x -q --timeout=400 -k "emulation or reproducer or graph or two_blocks or rebuilt" > $OUT/r05l_pytest.log 2>&1
tail -4 $OUT/r05l_pytest.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05l_bench.err | tail -1 > $OUT/r05l_bench.json
python - <<PY
import json
d = json.loads(open("$OUT/r05l_bench.json").read())
t = d["kernel_table"]
print(d["value"], d["ms_per_step"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"], "syncs/block", d["host_syncs_per_block"], "sum", t["sum_ms_per_block_without_symbol_ranking"])
PY
tail -3 $OUT/r05l_bench.err
timeout 120 python tools/dev/members_scale.py 8 8 2>&1 | tail -2
