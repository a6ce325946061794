#!/bin/bash
# One gpurun call of round 5 at HEAD: [the GPU test tier,] the round profile (PMC traffic passes, bench line with its members annex
# and per-kernel table, rocprofv3 kernel stats of the same command, ranking-chain timeline), members scaling, construction /
# command-line wall times.    bash tools/gpu_round5.sh <tag> [skip-tests]
set -u
TAG=${1:-r05}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1100 python -u -m pytest tests -m gpu -q --timeout=400 --durations=12 > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -18 $OUT/${TAG}_pytest_gpu.log
fi
timeout 600 bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_100MB_l1.json").read())
print({k: d.get(k) for k in ["value", "ms_per_step", "compressed_bytes", "compressed_sha256", "roundtrip_ok", "size_delta_pct", "stage_seconds_per_step", "host_syncs_per_block"]})
print(d.get("roofline"))
print(d.get("cpu_baseline"))
print(d.get("members"))
print(d.get("cpu_baseline_members"))
print("kernel table sum", d["kernel_table"]["sum_ms_per_block_without_symbol_ranking"])
PY
head -45 $OUT/${TAG}_bench_100MB_l1_kernel_stats.csv | cut -c1-150
cat $OUT/${TAG}_bench_timeline.jsonl | head -2
timeout 200 python tools/dev/members_scale.py 1 2 4 8 8 > $OUT/${TAG}_members_scale.jsonl 2>$OUT/${TAG}_members_scale.err
cat $OUT/${TAG}_members_scale.jsonl
timeout 90 python tools/dev/construct_time.py 2>&1 | grep -v "^encode:" > $OUT/${TAG}_construct_time.txt
python - <<PY
import sys
sys.path.insert(0, "tools")
import corpus
d = corpus.enwik_like(100_000_000)
open("/tmp/w100.bin", "wb").write(d)
with open("/tmp/w1g.bin", "wb") as f:
    for k in range(10):
        f.write(d)
PY
T0=$(date +%s.%N); bin/orz encode -s -l1 /tmp/w1g.bin /tmp/w1g.orz; T1=$(date +%s.%N)
echo "bin/orz encode -l1, 1,000,000,000 bytes: $(python -c "print(round($T1 - $T0, 3))") s wall = $(python -c "print(round(1000 / ($T1 - $T0), 1))") MB/s" >> $OUT/${TAG}_construct_time.txt
ls -l /tmp/w1g.orz | awk '{print "  -> " $5 " bytes"}' >> $OUT/${TAG}_construct_time.txt
cat $OUT/${TAG}_construct_time.txt
