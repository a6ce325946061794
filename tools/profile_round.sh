#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench JSON, rocprofv3 kernel stats of the same command,
# HBM traffic of the kernels from separate FETCH_SIZE / WRITE_SIZE passes on one 16 MiB block (fast mode).
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/one_block.py <<PY
import sys, os
repo = "$REPO"
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tools"))
import corpus, orz_amd
d = corpus.enwik_like(100_000_000)[:16 * 1024 * 1024]
print(len(orz_amd.encode_bytes(d, level=1)))
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $OUT/${TAG}_pmc_$c -- python /tmp/one_block.py > /dev/null 2>$OUT/${TAG}_pmc_$c.err
done
F=$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*_results.db' | head -1)
W=$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*_results.db' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $REPO/tools/pmc_summary.py $F $W $OUT/${TAG}_pmc_hbm_traffic_16MiB_l1.json
# the bench line below reports this run's traffic figures
cd $REPO
[ -s $OUT/${TAG}_pmc_hbm_traffic_16MiB_l1.json ] && cp $OUT/${TAG}_pmc_hbm_traffic_16MiB_l1.json $REPO/profiles/${TAG}_pmc_hbm_traffic_16MiB_l1.json
timeout 600 python bench.py --steps 3 --warmup 1 2>$OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_100MB_l1.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-members > $OUT/${TAG}_trace_bench.json 2>$OUT/${TAG}_trace.err
DB=$(find $OUT/${TAG}_trace -name '*_results.db' | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $OUT/${TAG}_bench_100MB_l1_kernel_stats.csv
# where the wall time goes: lead before the first ranking launch, gaps between the launches, time after the last one
[ -n "$DB" ] && python $REPO/tools/rocpd_timeline.py $DB > $OUT/${TAG}_bench_timeline.jsonl
# keep the merged output small: the databases stay on the box
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
ls -la $OUT | tail -12
