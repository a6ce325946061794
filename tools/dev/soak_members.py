"""dev: soak of the symbol-ranking guard -- eight encoders on one GPU encode members of the text workload for a given
number of seconds; every finished container is decoded by the library's host decoder and compared; the guard's messages
("... was repeated ...") arrive on stderr and are counted by the caller (tools/dev/soak_members.sh).
    python tools/dev/soak_members.py [seconds=120] [jobs=8]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus  # noqa: E402
import orz_amd  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
base = corpus.enwik_like(100_000_000)
member = 1 << 26
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
t0 = time.time()
members = blocks = rounds = bad = 0
first = None
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

pool = ThreadPoolExecutor(max_workers=3)  # the host decoder checks a container while the next ones are encoded
pending = []


def check(blob, n, data, rnd):
    try:
        back, nm = orz_amd.decode_members(blob)
        ok = nm == n and back == data
        why = "" if ok else "decodes to something else"
    except Exception as e:  # noqa: BLE001
        ok, why = False, str(e)
    if not ok:
        # which member?  (members are whole streams: decode them one by one)
        from orz_amd import dist as od
        pieces = od.split_members(blob)
        which = []
        for k, pc in enumerate(pieces):
            try:
                b, _ = orz_amd.decode_bytes(pc)
                if b != data[k * member:(k + 1) * member]:
                    which.append([k, "differs"])
            except Exception as e:  # noqa: BLE001
                which.append([k, str(e)[-60:]])
        sys.stderr.write("SOAK round %d: %s; members %r of %d\n" % (rnd, why, which, len(pieces)))
        try:  # the same member through a fresh single encoder: where do the two streams differ?
            import numpy as np
            k = which[0][0]
            ref = orz_amd.encode_bytes(data[k * member:(k + 1) * member], level=1)
            a, b = np.frombuffer(pieces[k], np.uint8), np.frombuffer(ref, np.uint8)
            m = min(len(a), len(b))
            d = np.nonzero(a[:m] != b[:m])[0]
            row = {"round": rnd, "member": k, "corpus_offset": (rnd * 7_919_113) % (len(base) - 1), "bad_bytes": len(a), "reference_bytes": len(b),
                   "reference_decodes": orz_amd.decode_bytes(ref)[0] == data[k * member:(k + 1) * member],
                   "differing_bytes_in_common_prefix": int(len(d)), "first_diff": int(d[0]) if len(d) else None, "last_diff": int(d[-1]) if len(d) else None,
                   "first_diffs": [[int(i), int(a[i]), int(b[i])] for i in d[:8]]}
            sys.stderr.write("SOAKDIFF " + json.dumps(row) + "\n")
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("SOAKDIFF failed: %r\n" % (e,))
    return ok


while time.time() - t0 < seconds:
    off = (rounds * 7_919_113) % (len(base) - 1)  # a different cut of the corpus every round
    data = (base[off:] + base[:off]) * ((jobs * member) // len(base) + 1)
    data = data[: jobs * member]
    blob, n = enc.encode(data, member_bytes=member)
    pending.append(pool.submit(check, blob, n, data, rounds))
    while len(pending) > 3:
        bad += 0 if pending.pop(0).result() else 1
    if first is None:
        first = len(blob)
    members += n
    blocks += n * 4
    rounds += 1
for f in pending:
    bad += 0 if f.result() else 1
enc.close()
print(json.dumps({"seconds": round(time.time() - t0, 1), "rounds": rounds, "members": members, "blocks": blocks, "bytes": rounds * jobs * member,
                  "containers_that_did_not_decode_to_their_input": bad, "first_container_bytes": first}))
