#!/bin/bash
# kernel-level profile of the fast mode on N bytes of the text workload: bash tools/gpu_fast_profile.sh <tag> [nbytes] [level]
set -u
TAG=${1:-fast}
N=${2:-33554432}
LV=${3:-1}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/fast_run.py <<PY
import sys, os, json
repo = "$REPO"
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tools"))
import corpus, orz_amd
d = corpus.zeros_noise($N) if os.environ.get("ORZ_PROFILE_ZEROS") else corpus.enwik_like($N)
enc = orz_amd.StreamEncoder(device=0, level=$LV, mode="fast")
enc.encode(d[:1000000])
out, st = enc.encode(d, stats=True)
print(json.dumps({"n": len(d), "out": len(out), "device_ms": st["total_ms"], "MBps": len(d) / 1e3 / st["total_ms"], "prep": st["t_prep_s"], "parse": st["t_parse_s"], "post": st["t_post_s"]}))
PY
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python /tmp/fast_run.py > $OUT/${TAG}_run.json 2>$OUT/${TAG}_trace.err
DB=$(find $OUT/${TAG}_trace -name '*_results.db' | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_trace
cat $OUT/${TAG}_run.json
cut -c1-150 $OUT/${TAG}_kernel_stats.csv | head -40
