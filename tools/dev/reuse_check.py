"""dev: a reused encoder must write what a fresh one writes.  Member B of the 1 GB text job through a fresh StreamEncoder, then
through encoders that encoded something else before; optionally through a MemberEncoder with one worker."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, orz_amd, _oracle
from orz_amd import dist as od
M = 1 << 26
data = corpus.enwik_like(1_000_000_000)
B = data[11 * M:12 * M]
enc = orz_amd.StreamEncoder(device=0, level=1); fresh = enc.encode(B); enc.close()
print("fresh", len(fresh), flush=True)
for name, first in [("member 3", data[3 * M:4 * M]), ("20 MB", data[:20_000_000]), ("3 MB", data[:3_000_000]), ("member 11 itself", B)]:
    enc = orz_amd.StreamEncoder(device=0, level=1)
    enc.encode(first)
    out = enc.encode(B)
    enc.close()
    print("after", name, len(out), "equal to fresh", out == fresh, "valid", _oracle.decode(out)[0] == B, flush=True)
me = orz_amd.MemberEncoder(device=0, level=1, jobs=1)
cont, nm = me.encode(data[10 * M:12 * M], member_bytes=M)
me.close()
p = od.split_members(cont)
print("MemberEncoder jobs=1, members 10+11: second", len(p[1]), "equal to fresh", p[1] == fresh, flush=True)
