"""dev: the per-block schedule (ORZ_FAST_SCHED) under eight encoders on one GPU -- aggregate MB/s and total size of 15 members of
64 MiB of the text workload at -l1 (input in HBM), every schedule twice, alternating; then one profiled single-encoder pass per
schedule: the sum of the non-ranking kernels per 16 MiB block."""
import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import torch
import corpus, orz_amd
base = corpus.enwik_like(100_000_000)
level = int(os.environ.get("SWEEP_LEVEL", "1"))
total = 15 << 26
data = (base * (total // len(base) + 1))[:total]
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
torch.cuda.synchronize()
scheds = sys.argv[1:] or ["0", "262144x3", "393216x3", "524288x3"]
for rep in range(2):
    for sc in scheds:
        os.environ["ORZ_FAST_SCHED"] = sc
        enc = orz_amd.MemberEncoder(device=0, level=level, jobs=8)
        enc.encode(data[: 8 << 20], member_bytes=1 << 20)
        t0 = time.time(); blob, n = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=1 << 26); t = time.time() - t0
        enc.close()
        print(json.dumps({"sched": sc, "rep": rep, "level": level, "MBps": round(len(data) / t / 1e6, 1), "bytes": len(blob)}), flush=True)
d100 = torch.frombuffer(bytearray(base), dtype=torch.uint8).to("cuda:0")
for sc in scheds:
    os.environ["ORZ_FAST_SCHED"] = sc
    enc = orz_amd.StreamEncoder(device=0, level=level)
    enc.encode_device(d100.data_ptr(), d100.numel())
    enc.set_profile(True)
    out, st = enc.encode_device(d100.data_ptr(), d100.numel(), stats=True)
    kt = enc.kernel_table()
    enc.close()
    blocks = len(base) / float(1 << 24)
    tot = sum(ms for name, ms, n in kt if "symrank" not in name) / blocks
    top = sorted(((round(ms / blocks, 2), name.split("<")[0][:28]) for name, ms, n in kt if "symrank" not in name), reverse=True)[:9]
    print(json.dumps({"sched": sc, "level": level, "sum_ms_per_block": round(tot, 2), "size": len(out), "syncs": st["host_syncs"], "top": top}), flush=True)
