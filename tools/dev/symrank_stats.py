"""dev: rank distribution of the hottest symbol-ranking context (oracle item trace of 8 MB of the workload, the context's table
replayed in Python: src/symrank.rs:61-97) -- what DESIGN.md 8b prices the no-move fast path with"""
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import _oracle, corpus
d = corpus.enwik_like(8_000_000)
o, tr = _oracle.encode(d, 1, trace_cap=4_000_000)
sym = np.array([t.symbol for t in tr], dtype=np.uint32); ctx = np.array([t.ctx for t in tr], dtype=np.uint32)
first = min(len(tr), 1 << 20)
cnt = np.bincount(sym[:first], minlength=389)
order = sorted(range(389), key=lambda s: (-max(int(cnt[s]), 1), s))
hot = np.bincount(ctx, minlength=512).argmax()
items = sym[ctx == hot]
print("items", len(tr), "hot ctx", hot, "items in it", len(items))
val = list(order); idx = [0]*389
for r, v in enumerate(val): idx[v] = r
c, s = 0, 1000000
hist = np.zeros(389, dtype=np.int64); moves0 = 0
for v in items.tolist():
    i = idx[v]; hist[i] += 1
    if c > 389: c = c*9//10; s = s*9//10
    c += 1; s += i
    ni = max(i - (i//16 + (s//16//c)), i//2, 0) if i - (i//16 + (s//16//c)) > 0 else i//2
    n = i - ni
    if n == 1:
        o1 = val[ni]; val[i] = o1; idx[o1] = i; val[ni] = v; idx[v] = ni
    elif n > 1:
        mid = ni + n//2; x = val[mid]; y = val[ni]
        val[i] = x; idx[x] = i; val[mid] = y; idx[y] = mid; val[ni] = v; idx[v] = ni
tot = hist.sum()
print("rank 0: %.3f  rank 1: %.3f  ranks 0..3: %.3f  0..15: %.3f  0..63: %.3f" % (hist[0]/tot, hist[1]/tot, hist[:4].sum()/tot, hist[:16].sum()/tot, hist[:64].sum()/tot))
