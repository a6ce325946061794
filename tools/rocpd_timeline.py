"""Where the wall time of an encode goes, from a rocprofv3 rocpd database (…_results.db, --kernel-trace): the
symbol-ranking launches are the serial chain of a stream, so the interesting numbers are the time before
the first one (the lead block's parse), the gaps between consecutive ones (the parse chain not keeping up), and the time
after the last one (tail stage, output hand-off).  One line per encode pass (passes are cut where more than `gap_ms`
pass without any kernel).

  python tools/rocpd_timeline.py <results.db> [kernel-substring=symrank] [gap_ms=20]
"""
import sqlite3
import sys


def passes(db_path, needle="symrank", gap_ms=20.0):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select d.start, d.end, k.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k "
        "on d.kernel_id = k.id order by d.start"
    ).fetchall()
    out, cur = [], []
    last_end = None
    for s, e, name in rows:
        if last_end is not None and s - last_end > gap_ms * 1e6 and cur:
            out.append(cur)
            cur = []
        cur.append((s, e, name))
        last_end = e if last_end is None else max(last_end, e)
    if cur:
        out.append(cur)
    res = []
    for p in out:
        chain = [(s, e) for s, e, n in p if needle in n]
        if not chain:
            continue
        t0, t1 = p[0][0], max(e for _, e, _ in p)
        gaps = [chain[i + 1][0] - chain[i][1] for i in range(len(chain) - 1)]
        # was the chain waiting for its input?  end of the last `ready` kernel (the per-context gather's run starts, the
        # last kernel before a ranking launch) before each chain launch, relative to that launch's start
        ready = sorted(e for _, e, n in p if "SymRunStart" in n)
        waits = []
        for s0, _ in chain:
            prev = [e for e in ready if e <= s0]
            waits.append(round((s0 - prev[-1]) / 1e6, 2) if prev else None)
        res.append({
            "kernels": len(p), "chain_launches": len(chain), "wall_ms": round((t1 - t0) / 1e6, 2),
            "lead_ms": round((chain[0][0] - t0) / 1e6, 2), "chain_busy_ms": round(sum(e - s for s, e in chain) / 1e6, 2),
            "gaps_ms": [round(g / 1e6, 2) for g in gaps], "after_last_ms": round((t1 - chain[-1][1]) / 1e6, 2),
            "input_ready_before_start_ms": waits,
        })
    return res


if __name__ == "__main__":
    import json

    needle = sys.argv[2] if len(sys.argv) > 2 else "symrank"
    gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    for r in passes(sys.argv[1], needle, gap):
        print(json.dumps(r))
