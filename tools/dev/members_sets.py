"""dev: why does a process's first set of eight encoders run at 530 MB/s and every later set at 410?  (a) one set reused three
times; (b) three sets created up front, used one after the other; (c) sets created and closed in turn (the slow pattern)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus, orz_amd
base = corpus.enwik_like(100_000_000)
jobs = 8
data = (base * 6)[: jobs * 64 * (1 << 20)]
def run(enc):
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)
    t0 = time.time(); blob, n = enc.encode(data, member_bytes=1 << 26); t1 = time.time()
    return round(len(data) / (t1 - t0) / 1e6, 1)
mode = sys.argv[1]
if mode == "d":
    import torch
    keep = [torch.cuda.Stream(device=0) for _ in range(64)]
    ext = [torch.cuda.ExternalStream]  # (kept alive)
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    print("64 streams created before the first set:", [run(enc) for _ in range(2)]); enc.close()
elif mode == "e":
    import torch
    keep = torch.empty(45 << 30, dtype=torch.uint8, device="cuda:0")
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    print("45 GB allocated before the first set:", [run(enc) for _ in range(2)]); enc.close()
elif mode == "a":
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    print("one set reused:", [run(enc) for _ in range(3)]); enc.close()
elif mode == "b":
    encs = [orz_amd.MemberEncoder(device=0, level=1, jobs=jobs) for _ in range(3)]
    print("three sets created up front:", [run(e) for e in encs])
    for e in encs: e.close()
else:
    out = []
    for _ in range(3):
        enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs); out.append(run(enc)); enc.close()
    print("sets created and closed in turn:", out)
