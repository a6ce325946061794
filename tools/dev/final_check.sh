#!/bin/bash
# dev: the driver's default bench command once, with every annex; prints what matters
export TMPDIR=/tmp
timeout 400 python bench.py > gpurun_out/r05_bench_check.json 2> gpurun_out/r05_bench_check.err; echo "rc $?"
python - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_check.json").read().strip().splitlines()[-1])
print(d["value"], d["roundtrip_ok"], d["size_delta_pct"], d["members"]["value"], d["cpu_baseline_members"]["value"], d["gpu_over_cpu_members"])
print(len(d["roofline_others"]), [(r["kernel"][-18:], r["device_ms_per_step"], r["traffic"] is not None) for r in d["roofline_others"]])
PY
tail -2 gpurun_out/r05_bench_check.err
