#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -u -m pytest tests/test_gpu_fast.py tests/test_gpu_verify.py tests/test_gpu_arena.py tests/test_gpu_soak.py -m gpu -x -q --timeout=400 \
  -k "not soak_rotations" > $OUT/r05h_pytest.log 2>&1
tail -5 $OUT/r05h_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-members > $OUT/r05h_bench.json 2> $OUT/r05h_bench.err
python - <<PY
import json
d = json.loads(open("$OUT/r05h_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ["value", "ms_per_step", "compressed_bytes", "roundtrip_ok", "size_delta_pct", "stage_seconds_per_step", "host_syncs_per_block"]})
print("sum", d["kernel_table"]["sum_ms_per_block_without_symbol_ranking"])
PY
for sh in 0 2 4; do
  echo "== ORZ_SHARE_MAIN=$sh"; ORZ_SHARE_MAIN=$sh timeout 120 python tools/dev/members_scale.py 8 8 2>&1 | tail -2
done
ORZ_SHARE_MAIN=2 timeout 120 python tools/dev/members_scale.py 1 2 4 8 12 16 2>&1 | tail -6
