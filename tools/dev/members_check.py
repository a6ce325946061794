"""dev: 1 GB of the text workload as 64 MiB members (8 encoders on one GPU), every member through the oracle's decoder;
prints which members fail.  python tools/dev/members_check.py [level=1] [repeats=2]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, orz_amd, _oracle
from orz_amd import dist as od
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
M = 1 << 26
data = corpus.enwik_like(1_000_000_000)
_oracle.lib()
def check(args):
    k, piece = args
    try:
        back, used = _oracle.decode(piece)
        return k, back == data[k * M:(k + 1) * M] and used == len(piece), len(piece)
    except Exception as e:
        return k, "invalid", len(piece)
for r in range(reps):
    enc = orz_amd.MemberEncoder(device=0, level=level, jobs=jobs)
    t0 = time.time(); container, nm = enc.encode(data, member_bytes=M); t = time.time() - t0
    enc.close()
    pieces = od.split_members(container)
    with ThreadPoolExecutor(max_workers=16) as ex:
        res = list(ex.map(check, enumerate(pieces)))
    bad = [(k, ok) for k, ok, _ in res if ok is not True]
    print({"rep": r, "level": level, "jobs": jobs, "members": nm, "MBps": round(len(data) / t / 1e6, 1), "sizes": [n for _, _, n in res], "bad": bad}, flush=True)
