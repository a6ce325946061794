#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dev/bisect_emu.py fewer-launches
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members 2>$OUT/r05m.err | tail -1 > $OUT/r05m.json
python - <<PY
import json
d = json.loads(open("$OUT/r05m.json").read())
t = d["kernel_table"]
print(d["value"], d["compressed_bytes"], d["roundtrip_ok"], d["stage_seconds_per_step"], "sum", t["sum_ms_per_block_without_symbol_ranking"])
for r in t["rows"][1:9]: print("    %-30s %7.1f %9.2f %8.3f" % (r["kernel"][:30], r["launches_per_block"], r["avg_launch_us"], r["ms_per_block"]))
PY
for f in 1 0 1 0; do echo "== fused=$f"; ORZ_FAST_FUSED=$f timeout 120 python tools/dev/members_scale.py 8 2>&1 | tail -1; done
