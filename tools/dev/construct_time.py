"""dev: what a process pays before and after the encode itself -- HIP start, encoder construction (allocations), the
first encode of a fresh encoder (graph capture), close -- and bin/orz's wall time on the 100 MB workload"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
t0 = time.time()
import corpus, orz_amd
t1 = time.time()
data = corpus.enwik_like(100_000_000)
t2 = time.time()
rows = []
try:  # device memory one stream encoder holds (hipMemGetInfo through torch, when it is there)
    import torch
    free0 = torch.cuda.mem_get_info(0)[0]
    e0 = orz_amd.StreamEncoder(device=0, level=1)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    e0.encode(data[:20_000_000])
    held_after = free0 - torch.cuda.mem_get_info(0)[0]
    e0.close()
    print({"device_bytes_held_by_one_stream_encoder": held, "after_encoding_20MB_of_text": held_after})
except Exception as ex:  # noqa: BLE001
    print({"device_bytes_held_by_one_stream_encoder": "n/a (%r)" % (ex,)})
for k in range(3):
    a = time.time(); enc = orz_amd.StreamEncoder(device=0, level=1); b = time.time()
    out = enc.encode(data[:20_000_000]); c = time.time()
    out = enc.encode(data[:20_000_000]); d = time.time()
    enc.close(); e = time.time()
    rows.append({"construct_ms": round((b - a) * 1e3, 1), "first_encode_20MB_ms": round((c - b) * 1e3, 1), "second_encode_20MB_ms": round((d - c) * 1e3, 1), "close_ms": round((e - d) * 1e3, 1)})
print({"import_s": round(t1 - t0, 2), "workload_s": round(t2 - t1, 2), "encoders": rows})
path = "/tmp/w100.bin"
open(path, "wb").write(data)
for lvl in ("-l1",):
    for k in range(3):
        a = time.time()
        subprocess.check_call([os.path.join(ROOT, "bin", "orz"), "encode", lvl, path, "/tmp/w100.orz"])
        b = time.time()
        print({"bin/orz encode": lvl, "wall_ms": round((b - a) * 1e3, 1), "MBps": round(len(data) / (b - a) / 1e6, 1), "out": os.path.getsize("/tmp/w100.orz")})
