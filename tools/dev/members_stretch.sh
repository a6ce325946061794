#!/bin/bash
# dev: per-kernel stats of J encoders sharing one GPU (rocprofv3 kernel trace of a members job) -> gpurun_out/<tag>_members_kernel_stats.csv
# (beside the solo stats of the bench command they show which kernels stretch under concurrency) + the concurrency summary
TAG=${1:-ms}; J=${2:-8}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/members_one.py <<PY
import sys, os, time
repo = "$REPO"
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tools"))
import corpus, orz_amd
J = $J
base = corpus.enwik_like(100_000_000)
data = (base * ((J * 64 * (1 << 20)) // len(base) + 1))[: J * 64 * (1 << 20)]
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=J)
enc.encode(data[: J * (1 << 20)], member_bytes=1 << 20)
time.sleep(0.5)
t0 = time.time(); blob, n = enc.encode(data, member_bytes=1 << 26); t = time.time() - t0
enc.close()
print({"jobs": J, "MBps": round(len(data) / t / 1e6, 1), "s": round(t, 3)})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/ms_$TAG -- python /tmp/members_one.py > $OUT/${TAG}_members_trace.log 2>&1
DB=$(find /tmp/ms_$TAG -name '*_results.db' | head -1)
tail -1 $OUT/${TAG}_members_trace.log
python $REPO/tools/rocpd_summary.py $DB > $OUT/${TAG}_members_kernel_stats.csv
python $REPO/tools/rocpd_concurrency.py $DB 100 symrank | tee $OUT/${TAG}_concurrency_nosymrank.json | cut -c1-600
