#!/bin/bash
# One gpurun call of round 3: GPU test tier, round profile (bench + kernel stats + PMC traffic), members scaling.
#   bash tools/gpu_round3.sh <tag> [skip-tests]
set -u
TAG=${1:-r03a}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -5 $OUT/${TAG}_pytest_gpu.log
fi
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
tail -3 $OUT/${TAG}_bench_100MB_l1.json | cut -c1-600
head -30 $OUT/${TAG}_bench_100MB_l1_kernel_stats.csv | cut -c1-150
timeout 600 python tools/dev/members_scale.py ${JOBS:-1 2 4 8} > $OUT/${TAG}_members_scale.jsonl 2>$OUT/${TAG}_members_scale.err
cat $OUT/${TAG}_members_scale.jsonl
