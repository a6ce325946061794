#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dev/bisect_emu.py flip-summary
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05k_$tag.err | tail -1 > $OUT/r05k_$tag.json
python - <<PY
import json
d = json.loads(open("$OUT/r05k_$tag.json").read())
t = d["kernel_table"]
print("$tag", d["value"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"], "syncs/block", d["host_syncs_per_block"], "sum", t["sum_ms_per_block_without_symbol_ranking"])
for r in t["rows"][1:12]: print("    %-30s %7.1f %9.2f %8.3f" % (r["kernel"][:30], r["launches_per_block"], r["avg_launch_us"], r["ms_per_block"]))
PY
}
run base ORZ_X=1
timeout 120 python tools/dev/members_scale.py 8 8 2>&1 | tail -2
