"""dev tool: item stream of one block (from the oracle's trace) as input for tools/dev/symrank_bench.hip"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _oracle  # noqa: E402
import corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
d = corpus.enwik_like(n)
o, tr = _oracle.encode(d, 1, trace_cap=4_000_000)
sym = np.array([t.symbol for t in tr], dtype=np.uint32)
unl = np.array([t.unlikely for t in tr], dtype=np.uint32)
ctx = np.array([t.ctx for t in tr], dtype=np.uint32)
rank = np.array([t.rank for t in tr], dtype=np.uint16)
first = min(len(tr), 1 << 20)
cnt = np.bincount(sym[:first], minlength=389)
order = sorted(range(389), key=lambda s: (-max(int(cnt[s]), 1), s))
perm = np.argsort(ctx, kind="stable")
gsym = (sym | (unl << 16))[perm].astype(np.uint32)
grank = rank[perm]
rstart = np.searchsorted(ctx[perm], np.arange(513)).astype(np.uint32)
state = np.zeros((512, 389 * 2 + 4), dtype=np.uint16)
for r, v in enumerate(order):
    state[:, r] = v
    state[:, 389 + v] = r
state[:, 389 * 2 + 0] = 0
state[:, 389 * 2 + 1] = 0
state[:, 389 * 2 + 2] = 1000000 & 0xffff
state[:, 389 * 2 + 3] = 1000000 >> 16
out = os.path.join(ROOT, "build", "symrank_case.bin")
with open(out, "wb") as f:
    f.write(np.array([len(gsym)], dtype=np.uint32).tobytes())
    f.write(gsym.tobytes()); f.write(rstart.tobytes()); f.write(state.tobytes()); f.write(grank.tobytes())
print("items", len(gsym), "hottest ctx", int(np.max(np.diff(rstart))), "->", out)
