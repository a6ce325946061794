"""dev: hunt for the GPU-only non-determinism of the members path.

Parent mode:   python tools/dev/hunt.py [plan.json]
    runs each configuration of the plan in a subprocess of its own (the library reads its environment knobs once), collects
    one JSON line per repetition -> gpurun_out/hunt.jsonl, and a summary per configuration: how many distinct streams each
    member came out as, which of them the oracle's decoder rejects, and what the forensic decoder (oracle/orz_diag.c) says
    about the first wrong item of each rejected one.  Rejected streams are kept (gpurun_out/hunt_bad_*.orz, at most three).
Worker mode:   python tools/dev/hunt.py --worker <round> <jobs> <reps> <fresh 0|1> <members>
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out")
MEMBER = 1 << 26


def soak_data(rnd, members):
    cache = "/tmp/hunt_r%d_%d.bin" % (rnd, members)
    if os.path.exists(cache):
        with open(cache, "rb") as f:
            return f.read()
    import corpus

    base = corpus.enwik_like(100_000_000)
    off = (rnd * 7_919_113) % (len(base) - 1)
    rot = base[off:] + base[:off]
    data = bytes((rot * ((members * MEMBER) // len(rot) + 1))[: members * MEMBER])
    with open(cache + ".tmp", "wb") as f:
        f.write(data)
    os.replace(cache + ".tmp", cache)
    return data


def worker(rnd, jobs, reps, fresh, members):
    import _oracle
    import orz_amd
    from orz_amd import dist as od
    from concurrent.futures import ThreadPoolExecutor

    data = soak_data(rnd, members)
    seen = {}  # (member, sha) -> bytes
    enc = None
    for rep in range(reps):
        if enc is None or fresh:
            if enc is not None:
                enc.close()
            enc = orz_amd.MemberEncoder(device=0, level=1, jobs=jobs)
        t0 = time.time()
        try:
            blob, nm = enc.encode(data, member_bytes=MEMBER)
        except Exception as e:  # noqa: BLE001  (the product's own verifier may fail the encode)
            print(json.dumps({"rep": rep, "error": str(e)[-300:]}), flush=True)
            enc.close()
            enc = None
            continue
        dt = time.time() - t0
        pieces = od.split_members(blob)
        shas = []
        for k, pc in enumerate(pieces):
            h = hashlib.sha256(pc).hexdigest()[:16]
            shas.append(h)
            if (k, h) not in seen:
                seen[(k, h)] = pc
        print(json.dumps({"rep": rep, "s": round(dt, 2), "sizes": [len(p) for p in pieces], "sha": shas}), flush=True)
    if enc is not None:
        enc.close()

    def check(item):
        (k, h), pc = item
        exp = data[k * MEMBER:(k + 1) * MEMBER]
        d = _oracle.diag(pc, exp)
        return k, h, len(pc), d

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
        res = list(ex.map(check, list(seen.items())))
    kept = len([f for f in os.listdir(OUT) if f.startswith("hunt_bad_")]) if os.path.isdir(OUT) else 0
    for k, h, n, d in res:
        row = {"variant": [k, h], "bytes": n, "valid": not d}
        if d:
            row["diag"] = d
            if kept < 3:
                try:
                    with open(os.path.join(OUT, "hunt_bad_r%d_m%d_%s.orz" % (rnd, k, h)), "wb") as f:
                        f.write(seen[(k, h)])
                    kept += 1
                except OSError:
                    pass
        print(json.dumps(row), flush=True)


DEFAULT_PLAN = [
    {"name": "default jobs=8 fresh", "env": {}, "round": 40, "jobs": 8, "reps": 24, "fresh": 1},
    {"name": "graphs off", "env": {"ORZ_GRAPHS": "0"}, "round": 40, "jobs": 8, "reps": 24, "fresh": 1},
    {"name": "jobs=1 (one encoder, members in turn)", "env": {}, "round": 40, "jobs": 1, "reps": 6, "fresh": 1},
    {"name": "jobs=8 reused encoders", "env": {}, "round": 40, "jobs": 8, "reps": 16, "fresh": 0},
    {"name": "verify off", "env": {"ORZ_FAST_VERIFY": "0"}, "round": 40, "jobs": 8, "reps": 12, "fresh": 1},
    {"name": "copy stream off", "env": {"ORZ_COPY_STREAM": "0"}, "round": 40, "jobs": 8, "reps": 12, "fresh": 1},
]


def main():
    os.makedirs(OUT, exist_ok=True)
    plan = DEFAULT_PLAN
    if len(sys.argv) > 1:
        with open(sys.argv[1]) as f:
            plan = json.load(f)
    log = open(os.path.join(OUT, "hunt.jsonl"), "a")
    for cfg in plan:
        env = dict(os.environ)
        env.update(cfg.get("env", {}))
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(cfg["round"]), str(cfg["jobs"]), str(cfg["reps"]),
                            str(cfg.get("fresh", 1)), str(cfg.get("members", 8))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=cfg.get("timeout", 900))
        rows = []
        for ln in p.stdout.splitlines():
            try:
                rows.append(json.loads(ln))
            except ValueError:
                pass
        reps = [r for r in rows if "sha" in r]
        errs = [r for r in rows if "error" in r]
        variants = [r for r in rows if "variant" in r]
        nm = max((len(r["sha"]) for r in reps), default=0)
        distinct = [len({r["sha"][k] for r in reps}) for k in range(nm)]
        counts = []
        for k in range(nm):
            c = {}
            for r in reps:
                c[r["sha"][k]] = c.get(r["sha"][k], 0) + 1
            counts.append(c)
        summary = {"config": cfg["name"], "env": cfg.get("env", {}), "round": cfg["round"], "jobs": cfg["jobs"], "reps": len(reps),
                   "errors": errs[:4], "n_errors": len(errs), "rc": p.returncode, "seconds": round(time.time() - t0, 1),
                   "distinct_streams_per_member": distinct, "counts": counts,
                   "invalid": [v for v in variants if not v["valid"]],
                   "stderr_tail": p.stderr[-1500:], "avg_encode_s": round(sum(r["s"] for r in reps) / max(1, len(reps)), 2)}
        log.write(json.dumps(summary) + "\n")
        log.flush()
        print(json.dumps({k: summary[k] for k in ("config", "reps", "n_errors", "rc", "seconds", "distinct_streams_per_member", "avg_encode_s")}
                         | {"n_invalid": len(summary["invalid"])}), flush=True)
        for v in summary["invalid"]:
            print("   INVALID", json.dumps(v), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) != 0, int(sys.argv[6]))
    else:
        main()
