#!/bin/bash
# One gpurun call of round 6 at HEAD: [the GPU test tier,] the round profile (PMC traffic passes, bench line with its annexes and
# per-kernel table, rocprofv3 kernel stats of the same command, ranking-chain timeline), an eight-encoder kernel trace with the
# input in HBM (how much of the wall time the GPU has nothing in flight), members scaling, the device decoder's rate,
# construction / command-line wall times.    bash tools/gpu_round6.sh <tag> [skip-tests]
set -u
TAG=${1:-r06}
OUT=gpurun_out
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1300 python -u -m pytest tests -m gpu -q --timeout=500 --durations=12 > $OUT/${TAG}_pytest_gpu.log 2>&1
  tail -18 $OUT/${TAG}_pytest_gpu.log
fi
timeout 700 bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_100MB_l1.json").read())
print({k: d.get(k) for k in ["value", "ms_per_step", "compressed_bytes", "compressed_sha256", "roundtrip_ok", "size_delta_pct", "stage_seconds_per_step", "host_syncs_per_block"]})
print(d.get("roofline"))
print(d.get("cpu_baseline"))
for k in ("members", "members_l2_text", "members_l2_zeros"):
    m = d.get(k) or {}
    print(k, {x: m.get(x) for x in ("value", "size_delta_pct", "roundtrip_ok", "error", "gpu_over_cpu_members")})
print(d.get("cpu_baseline_members"))
print("kernel table sum", d["kernel_table"]["sum_ms_per_block_without_symbol_ranking"])
PY
head -45 $OUT/${TAG}_bench_100MB_l1_kernel_stats.csv | cut -c1-150
cat $OUT/${TAG}_bench_timeline.jsonl | head -2
# ---- eight encoders, input in HBM, 15 members of 64 MiB: kernel trace -> time with nothing in flight
cat > /tmp/members_hbm.py <<PY
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
import torch, corpus, orz_amd
base = corpus.enwik_like(100_000_000)
total = 15 << 26
data = (base * (total // len(base) + 1))[:total]
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0"); torch.cuda.synchronize()
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
enc.encode(data[: 8 << 20], member_bytes=1 << 20)
time.sleep(0.5)
t0 = time.time(); blob, n = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=1 << 26); t = time.time() - t0
enc.close()
print({"jobs": 8, "members": n, "MBps": round(len(data) / t / 1e6, 1), "s": round(t, 3)})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/mt_$TAG -- python /tmp/members_hbm.py > $REPO/$OUT/${TAG}_members8_trace.log 2>&1
DB=$(find /tmp/mt_$TAG -name '*_results.db' | head -1)
cd $REPO
tail -1 $OUT/${TAG}_members8_trace.log
[ -n "$DB" ] && python tools/rocpd_concurrency.py $DB 100 > $OUT/${TAG}_members8_concurrency.json && cut -c1-600 $OUT/${TAG}_members8_concurrency.json
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${TAG}_members8_kernel_stats.csv
rm -rf /tmp/mt_$TAG
timeout 200 python tools/dev/members_scale.py 1 2 4 8 8 > $OUT/${TAG}_members_scale.jsonl 2>$OUT/${TAG}_members_scale.err
cat $OUT/${TAG}_members_scale.jsonl
timeout 200 python tools/dev/decode_bench.py 32 4 2>/dev/null | tail -1 > $OUT/${TAG}_decode_bench.jsonl; cat $OUT/${TAG}_decode_bench.jsonl
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
