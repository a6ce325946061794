"""dev: aggregate encode rate of J concurrent stream encoders on one GPU (members of 64 MiB, text workload, -l1)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus, orz_amd
base = corpus.enwik_like(100_000_000)
for jobs in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    data = (base * ((jobs * 64 * (1 << 20)) // len(base) + 1))[: jobs * 64 * (1 << 20)]
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)
    t0 = time.time(); blob, n = enc.encode(data, member_bytes=1 << 26); t = time.time() - t0
    enc.close()
    print(json.dumps({"jobs": jobs, "bytes": len(data), "members": n, "MBps": round(len(data) / t / 1e6, 1), "s": round(t, 2), "ratio": round(len(blob) / len(data), 5)}), flush=True)
