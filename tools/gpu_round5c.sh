#!/bin/bash
# experiments: what the scans inside FastEval cost (time only; ORZ_FAST_DBG bits 4 / 8 / 16, ORZ_FAST_NEAR)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05c_$tag.err | tail -1 > $OUT/r05c_$tag.json
python - <<PY
import json
d = json.loads(open("$OUT/r05c_$tag.json").read())
r = [d["roofline"]] + d["roofline_others"]
print("$tag", d["value"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"], [(x["kernel"][-14:], x["avg_launch_us"], x["device_ms_per_step"]) for x in r])
PY
}
run base ORZ_X=1
run dbg16 ORZ_FAST_DBG=16
run dbg4 ORZ_FAST_DBG=4
run dbg8 ORZ_FAST_DBG=8
run near0 ORZ_FAST_NEAR=0
run extra0 ORZ_FAST_EXTRA=0
