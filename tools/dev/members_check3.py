"""dev: tests/test_gpu_configs.py's C2 case exactly as the test runs it (oracle work included), then the C3 encode with
every member checked and failing members dumped"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, orz_amd, _oracle
import test_gpu_configs as T
from orz_amd import dist as od
M = 1 << 26
_oracle.lib()
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
data = corpus.enwik_like(1_000_000_000)
if mode == "full":
    T._members_case(_oracle, "C2 (dev)", data, 2, True)
elif mode == "sleep":
    enc = orz_amd.MemberEncoder(device=0, level=2, jobs=8); enc.encode(data, member_bytes=M); enc.close(); time.sleep(60)
def check(args):
    k, piece = args
    try:
        back, used = _oracle.decode(piece)
        return k, back == data[k * M:(k + 1) * M] and used == len(piece)
    except Exception:
        return k, "invalid"
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
container, nm = enc.encode(data, member_bytes=M)
enc.close()
pieces = od.split_members(container)
with ThreadPoolExecutor(max_workers=16) as ex:
    res = list(ex.map(check, enumerate(pieces)))
bad = [(k, ok) for k, ok in res if ok is not True]
print({"mode": mode, "members": nm, "sizes": [len(p) for p in pieces], "bad": bad}, flush=True)
for k, _ in bad[:2]:
    open(os.path.join(ROOT, "gpurun_out", "r03o_bad_member_%d.orz" % k), "wb").write(pieces[k])
