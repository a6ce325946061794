"""dev: a multi-block stream encoded in units of a block (ORZ_FAST_UNIT) on the GPU decodes with the library's host decoder
and stays next to the whole-block encoding; plus the exact-mode bench line"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus, orz_amd
d = corpus.enwik_like(100_000_000)[:40_000_000]
res = {}
for unit in (0, 8 << 20, 4 << 20):
    if unit: os.environ["ORZ_FAST_UNIT"] = str(unit)
    enc = orz_amd.StreamEncoder(device=0, level=1)
    out = enc.encode(d); enc.close()
    back = orz_amd.decode_bytes(out)
    back = back[0] if isinstance(back, tuple) else back
    res[unit] = (len(out), hashlib.sha256(back).hexdigest() == hashlib.sha256(d).hexdigest())
print(json.dumps({str(k): v for k, v in res.items()}))
