"""dev: eight against twelve encoders on one GPU, input in host memory (MemberEncoder.encode) and in HBM (encode_device), -l1"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, corpus, orz_amd
base = corpus.enwik_like(100_000_000)
for jobs, nm in [(8, 16), (12, 24), (8, 16), (12, 24), (12, 12), (8, 8)]:
    total = nm << 26
    data = (base * (total // len(base) + 1))[:total]
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)
    t0 = time.time(); blob, n = enc.encode(data, member_bytes=1 << 26); th = time.time() - t0
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0"); torch.cuda.synchronize()
    t0 = time.time(); blob2, n2 = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=1 << 26); td = time.time() - t0
    enc.close(); del src
    print(json.dumps({"jobs": jobs, "members": n, "host_MBps": round(total / th / 1e6, 1), "device_MBps": round(total / td / 1e6, 1), "same": blob == blob2}), flush=True)
