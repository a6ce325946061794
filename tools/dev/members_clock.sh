#!/bin/bash
# dev: three sets of eight encoders in one process with the shader / memory clocks and the package power sampled beside them --
# is "a process's first set runs at 530 MB/s, every later one at 410" a clock question?
OUT=gpurun_out; mkdir -p $OUT
( for i in $(seq 1 700); do echo "$(date +%s.%N) $(rocm-smi -c -P 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.02; done ) > $OUT/r04_members_clock_samples.txt &
SP=$!
python - <<'PY'
import json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import corpus, orz_amd
base = corpus.enwik_like(100_000_000)
jobs = 8
data = (base * 6)[: jobs * 64 * (1 << 20)]
for k in range(3):
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)
    t0 = time.time(); blob, n = enc.encode(data, member_bytes=1 << 26); t1 = time.time()
    enc.close()
    print(json.dumps({"set": k, "t0": t0, "t1": t1, "MBps": round(len(data) / (t1 - t0) / 1e6, 1)}), flush=True)
    if k == 1:
        time.sleep(3.0)  # an idle pause before the third set
PY
kill $SP 2>/dev/null; wait $SP 2>/dev/null
echo samples: $(wc -l < $OUT/r04_members_clock_samples.txt)
