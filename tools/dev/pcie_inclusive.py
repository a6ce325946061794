"""dev: the bench workload handed over as a HOST buffer (page-locked for the call, block uploads by hipMemcpyAsync) -- the
PCIe-inclusive rate DESIGN.md quotes next to bench.py's HBM-resident one"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus, orz_amd
d = corpus.enwik_like(100_000_000)
enc = orz_amd.StreamEncoder(device=0, level=1)
enc.encode(d[:20_000_000])
best = 1e9
for _ in range(3):
    t0 = time.time(); out = enc.encode(d); best = min(best, time.time() - t0)
print(json.dumps({"host_input_MBps": round(len(d) / best / 1e6, 1), "ms": round(best * 1e3, 1), "out": len(out)}))
