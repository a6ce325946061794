"""dev: members k of the 1 GB text job (64 MiB each) encoded as single streams on a fresh encoder -> gpurun_out/good_member_k.orz
(the fast mode is deterministic: these are the bytes orz_members_encode must produce for the same members)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, orz_amd, _oracle
M = 1 << 26
data = corpus.enwik_like(1_000_000_000)
for k in [int(x) for x in sys.argv[1:]]:
    enc = orz_amd.StreamEncoder(device=0, level=1)
    out = enc.encode(data[k * M:(k + 1) * M])
    enc.close()
    ok = _oracle.decode(out)[0] == data[k * M:(k + 1) * M]
    open(os.path.join(ROOT, "gpurun_out", "good_member_%d.orz" % k), "wb").write(out)
    print("member", k, len(out), "oracle round trip", ok, flush=True)
