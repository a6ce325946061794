#!/bin/bash
# Round-end artefacts on the GPU box (run through gpurun):  bash tools/final_round.sh <tag>  -> gpurun_out/<tag>_*
# GPU tests, bench + rocprofv3 kernel stats + PMC traffic (tools/profile_round.sh), BASELINE configs[2]/[4] at 1 GB,
# the benchmark-tool rows of this repo and the oracle, members scaling on one GPU.
set -u
TAG=${1:-rXX}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/${TAG}_pytest_gpu.log
cat $OUT/${TAG}_pytest_gpu.log
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
cat $OUT/${TAG}_bench_100MB_l1.json
timeout 200 python tools/gpu_configs.py 1000000000 8 67108864 2 > $OUT/${TAG}_configs_c2_c4_1GB_l2.jsonl 2>$OUT/${TAG}_configs.err
cat $OUT/${TAG}_configs_c2_c4_1GB_l2.jsonl
timeout 150 python tools/benchmark_tool.py --corpus 100000000 --rounds 2 --skip-others --json $OUT/${TAG}_benchmark_tool.json > $OUT/${TAG}_benchmark_tool.md 2>$OUT/${TAG}_benchmark_tool.err
cat $OUT/${TAG}_benchmark_tool.md
timeout 120 python tools/dev/members_scale.py 1 2 4 8 > $OUT/${TAG}_members_scale.jsonl 2>/dev/null
cat $OUT/${TAG}_members_scale.jsonl
