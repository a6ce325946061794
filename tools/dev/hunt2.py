"""dev: the first 17 MiB of member 5 of soak round 40, eight copies through eight encoders, with the fast parse's state around the
position where the good and the bad stream part dumped after the rounds and after the repairs (ORZ_DEBUG_DUMP)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import hunt
OUT = os.path.join(ROOT, "gpurun_out")
dump = os.path.join(OUT, "dump")
os.makedirs(dump, exist_ok=True)
os.environ["ORZ_DEBUG_DUMP"] = dump
os.environ["ORZ_DEBUG_POS"] = sys.argv[1] if len(sys.argv) > 1 else "18623744"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
import orz_amd
from orz_amd import dist as od
M = 1 << 26
N = 17825792
piece = hunt.soak_data(40, 8)[5 * M:5 * M + N]
data = piece * 8
for rep in range(reps):
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
    blob, nm = enc.encode(data, member_bytes=N)
    enc.close()
    pieces = od.split_members(blob)
    print(json.dumps({"rep": rep, "sha": [hashlib.sha256(p).hexdigest()[:12] for p in pieces], "sizes": [len(p) for p in pieces]}), flush=True)
