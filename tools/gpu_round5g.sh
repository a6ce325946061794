#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05g_$tag.err | tail -1 > $OUT/r05g_$tag.json
python - <<PY
import json
d = json.loads(open("$OUT/r05g_$tag.json").read())
t = d["kernel_table"]
print("$tag", d["value"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"], "sum", t["sum_ms_per_block_without_symbol_ranking"], [(r["kernel"][:12], r["launches_per_block"], r["avg_launch_us"]) for r in t["rows"] if r["kernel"].startswith("FastSourceL")])
PY
}
run cap256 ORZ_X=1
run cap64 ORZ_FAST_SRCCAP=64
run cap32 ORZ_FAST_SRCCAP=32
run cap16 ORZ_FAST_SRCCAP=16
run cap8 ORZ_FAST_SRCCAP=8
