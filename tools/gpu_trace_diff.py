"""Find the first item where the GPU parse and the oracle's parse differ (run through gpurun)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
import corpus  # noqa: E402
import orz_amd  # noqa: E402

n = int(sys.argv[1]); level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
data = corpus.text_corpus(max(n, 1 << 20))[:n]
ref, items = _oracle.encode(data, level, trace_cap=n + 16)
enc = orz_amd.StreamEncoder(0, level, mode="exact")
enc.set_item_trace(True)
out = enc.encode(data)
tr = enc.item_trace()
print("equal", out == ref, "items gpu", len(tr), "oracle", len(items))
P = 16777215
# oracle positions are window offsets per block as well
k = 0
for i in range(min(len(tr), len(items))):
    o = items[i]; g = tr[i]
    if (g["pos"], g["symbol"], g["rank"], g["ctx"], g["enc_len"], g["unlikely"]) != (o.pos, o.symbol, o.rank, o.ctx, o.enc_len, o.unlikely):
        print("first diff at item", i, "block", g["block"])
        for j in range(max(0, i - 3), min(len(tr), i + 6)):
            o = items[j]; g = tr[j]
            print(j, "GPU pos=%d sym=%d rank=%d ctx=%d enc=%d unl=%d rob=%d/%d al=%d" % (g["pos"], g["symbol"], g["rank"], g["ctx"], g["enc_len"], g["unlikely"], g["robits"] & 0xfff, g["robits"] >> 12, g["after_literal"]),
                  "| ORC pos=%d sym=%d rank=%d ctx=%d enc=%d unl=%d ro=%d len=%d al=%d" % (o.pos, o.symbol, o.rank, o.ctx, o.enc_len, o.unlikely, o.reduced_offset, o.match_len, o.after_literal))
        k += 1
        break
if not k:
    print("no item difference in the common prefix")
