"""dev: ORZ_FAST_DBG experiment bits in ONE call -- per setting: size of the 100 MB text workload at -l1 / -l2 and of 100 MB of zeros
with noise at -l2 against the oracle, the single-encoder kernel sum per block and FastEval's share, eight encoders' MB/s."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import torch, corpus, orz_amd
text = corpus.enwik_like(100_000_000); zeros = corpus.zeros_noise(100_000_000)
res = {"dbg": os.environ.get("ORZ_FAST_DBG", "0")}
for name, d, lvl in (("text_l1", text, 1), ("text_l2", text, 2), ("zeros_l2", zeros, 2)):
    e = orz_amd.StreamEncoder(device=0, level=lvl)
    res[name] = len(e.encode(d)); e.close()
d100 = torch.frombuffer(bytearray(text), dtype=torch.uint8).to("cuda:0")
e = orz_amd.StreamEncoder(device=0, level=1)
e.encode_device(d100.data_ptr(), d100.numel())
e.set_profile(True)
out, st = e.encode_device(d100.data_ptr(), d100.numel(), stats=True)
kt = e.kernel_table(); e.close()
blocks = len(text) / float(1 << 24)
res["sum_ms_per_block"] = round(sum(ms for name, ms, n in kt if "symrank" not in name) / blocks, 2)
res["FastEval_ms"] = round(sum(ms for name, ms, n in kt if "FastEval" in name) / blocks, 2)
total = 15 << 26
data = (text * (total // len(text) + 1))[:total]
src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0"); torch.cuda.synchronize()
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
enc.encode(data[: 8 << 20], member_bytes=1 << 20)
res["members_MBps"] = []
for rep in range(2):
    t0 = time.time(); blob, n = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=1 << 26); t = time.time() - t0
    res["members_MBps"].append(round(len(data) / t / 1e6, 1))
enc.close()
print(json.dumps(res), flush=True)
''' % (ROOT, ROOT)
REF = {"text_l1": 28211999, "text_l2": 27749523}
for dbg in sys.argv[1:] or ["0", "128", "256"]:
    env = dict(os.environ, ORZ_FAST_DBG=dbg)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"dbg": dbg, "error": r.stderr[-600:]})
    try:
        d = json.loads(line)
        for k, ref in REF.items():
            if k in d:
                d[k + "_delta_pct"] = round(100.0 * (d[k] - ref) / ref, 4)
        line = json.dumps(d)
    except Exception:
        pass
    print(line, flush=True)
