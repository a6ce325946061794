#!/bin/bash
# One gpurun call of round 3: GPU test tier, round profile (bench + kernel stats + PMC traffic), members scaling,
# optional bench variants.
#   bash tools/gpu_round3.sh <tag> [skip-tests]      env: PYTEST_ARGS, JOBS, VARIANTS="NAME=VAL,NAME=VAL;NAME=VAL"
set -u
TAG=${1:-r03a}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  eval "timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-}" > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -5 $OUT/${TAG}_pytest_gpu.log
fi
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_100MB_l1.json").read())
print({k: d.get(k) for k in ["value", "ms_per_step", "compressed_bytes", "roundtrip_ok", "size_delta_pct", "stage_seconds_per_step"]})
PY
head -32 $OUT/${TAG}_bench_100MB_l1_kernel_stats.csv | cut -c1-150
cat $OUT/${TAG}_bench_timeline.jsonl | head -2
timeout 600 python tools/dev/members_scale.py ${JOBS:-1 2 4 8} > $OUT/${TAG}_members_scale.jsonl 2>$OUT/${TAG}_members_scale.err
cat $OUT/${TAG}_members_scale.jsonl
# bench variants: semicolon-separated sets of comma-separated NAME=VALUE
IFS=';' read -ra SETS <<< "${VARIANTS:-}"
for set in "${SETS[@]}"; do
  [ -z "$set" ] && continue
  ( IFS=','; for kv in $set; do export "$kv"; done
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>>$OUT/${TAG}_variants.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'variant': '$set', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'compressed_bytes': d['compressed_bytes'], 'roundtrip_ok': d['roundtrip_ok'], 'stage': d['stage_seconds_per_step']}))" ) | tee -a $OUT/${TAG}_variants.jsonl
done
