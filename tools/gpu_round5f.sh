#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05f_$tag.err | tail -1 > $OUT/r05f_$tag.json
python - <<PY
import json
d = json.loads(open("$OUT/r05f_$tag.json").read())
print("$tag", d["value"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"])
t = d["kernel_table"]
print("  sum", t["sum_ms_per_block_without_symbol_ranking"])
if "$tag" == "base":
    for r in t["rows"][:34]: print("  %-44s %7.1f %9.2f %8.3f" % (r["kernel"][:44], r["launches_per_block"], r["avg_launch_us"], r["ms_per_block"]))
else:
    print("  ", [(r["kernel"][:12], r["avg_launch_us"]) for r in t["rows"][:8]])
PY
}
run base ORZ_X=1
run noatomic ORZ_FAST_DBG=1024
timeout 120 python tools/dev/members_scale.py 1 8 8 2>&1 | tail -3
