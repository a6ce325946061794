"""dev: A/B of library builds (ORZ_LIB_PATH) in ONE gpurun call -- for each build, a subprocess that runs eight encoders over 15
members of 64 MiB of the text workload (input in HBM) twice and one profiled single-encoder pass (kernel sums per 16 MiB block).
    python tools/dev/lib_ab.py <lib.so> [<lib.so> ...]      (a '-' runs the product library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import torch, corpus, orz_amd
base = corpus.enwik_like(100_000_000)
level = int(os.environ.get("AB_LEVEL", "1"))
total = 15 << 26
data = (base * (total // len(base) + 1))[:total]
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0"); torch.cuda.synchronize()
res = {"lib": os.environ.get("ORZ_LIB_PATH", "product"), "members_MBps": []}
enc = orz_amd.MemberEncoder(device=0, level=level, jobs=8)
enc.encode(data[: 8 << 20], member_bytes=1 << 20)
for rep in range(3):
    t0 = time.time(); blob, n = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=1 << 26); t = time.time() - t0
    res["members_MBps"].append(round(len(data) / t / 1e6, 1)); res["members_bytes"] = len(blob)
enc.close(); del src
d100 = torch.frombuffer(bytearray(base), dtype=torch.uint8).to("cuda:0")
e = orz_amd.StreamEncoder(device=0, level=level)
e.encode_device(d100.data_ptr(), d100.numel())
out, st = e.encode_device(d100.data_ptr(), d100.numel(), stats=True)
res["stream_MBps"] = round(len(base) / st["total_ms"] / 1e3, 1); res["syncs"] = st["host_syncs"]; res["size"] = len(out)
e.set_profile(True)
out, st = e.encode_device(d100.data_ptr(), d100.numel(), stats=True)
kt = e.kernel_table(); e.close()
blocks = len(base) / float(1 << 24)
res["sum_ms_per_block"] = round(sum(ms for name, ms, n in kt if "symrank" not in name) / blocks, 2)
res["top"] = sorted(((round(ms / blocks, 2), name.split("<")[0][:24]) for name, ms, n in kt if "symrank" not in name), reverse=True)[:12]
print(json.dumps(res), flush=True)
''' % (ROOT, ROOT)
for lib in sys.argv[1:] or ["-"]:
    env = dict(os.environ)
    if lib != "-":
        env["ORZ_LIB_PATH"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"lib": lib, "error": r.stderr[-800:]}), flush=True)
