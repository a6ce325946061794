import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import corpus, orz_amd
d = corpus.enwik_like(100_000_000)
for lvl in (1, 2):
    enc = orz_amd.StreamEncoder(device=0, level=lvl)
    out, st = enc.encode(d, stats=True)
    print("level", lvl, len(out), st["host_syncs"], st["blocks"], flush=True)
    enc.close()
