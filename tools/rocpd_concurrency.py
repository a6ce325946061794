"""How well do the kernels of several encoders share one GPU?  From a rocprofv3 rocpd database (…_results.db,
--kernel-trace): for the busiest window of the trace, the wall time, the time at least one kernel was running (union),
the sum of the kernel durations, the time spent at each concurrency level, and the same split per hardware queue / stream
when the database names them.  "sum / union" is the average number of kernels in flight while the GPU was busy;
"union / wall" is how much of the wall time the GPU had anything to do at all (the rest: the host not keeping up, or
waiting for copies).

The trace is cut where the GPU idles for more than `gap_ms` (default 100: warm-up | sleep | the run that matters) and the
piece with the most kernel time is analysed.

  python tools/rocpd_concurrency.py <results.db> [gap_ms=100] [exclude-kernel-substring]
(e.g. exclude `symrank`: one launch of it lasts a whole block and hides whether the parse kernels overlap)
"""
import json
import sqlite3
import sys


def columns(db, table):
    return [r[1] for r in db.execute("pragma table_info(%s)" % table).fetchall()]


def analyse(db_path, gap_ms=100.0, exclude=None):
    db = sqlite3.connect(db_path)
    cols = columns(db, "rocpd_kernel_dispatch")
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "select d.start, d.end, k.kernel_name%s%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start" % (
        ", d.%s" % qcol if qcol else ", 0", ", d.%s" % scol if scol else ", 0")
    rows = db.execute(sel).fetchall()
    if not rows:
        return {}
    pieces, cur, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[0] - last_end > gap_ms * 1e6 and cur:
            pieces.append(cur)
            cur = []
        cur.append(r)
        last_end = r[1] if last_end is None else max(last_end, r[1])
    pieces.append(cur)
    rows = max(pieces, key=lambda p: sum(r[1] - r[0] for r in p))
    if exclude:
        rows = [r for r in rows if exclude not in r[2]]
    ev = []
    for s, e, _, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, at_level = 0, ev[0][0], {}
    for t, d in ev:
        if t > last:
            at_level[level] = at_level.get(level, 0) + (t - last)
            last = t
        level += d
    wall = ev[-1][0] - ev[0][0]
    union = sum(v for k, v in at_level.items() if k > 0)
    total = sum(e - s for s, e, _, _, _ in rows)
    by_q, by_s, by_k = {}, {}, {}
    for s, e, n, q, st in rows:
        by_q[q] = by_q.get(q, 0) + (e - s)
        by_s[st] = by_s.get(st, 0) + (e - s)
        short = n.split("INS_")[-1].split("EEEv")[0] if "INS_" in n else n[:40]
        a = by_k.setdefault(short, [0, 0])
        a[0] += 1
        a[1] += e - s
    lv = sorted(at_level.items())
    return {
        "kernels": len(rows), "wall_ms": round(wall / 1e6, 2), "busy_union_ms": round(union / 1e6, 2), "sum_kernel_ms": round(total / 1e6, 2),
        "avg_in_flight_while_busy": round(total / union, 2) if union else None, "busy_frac_of_wall": round(union / wall, 3) if wall else None,
        "ms_at_concurrency": {str(k): round(v / 1e6, 2) for k, v in lv if v > 0.005e6},
        "queues": len(by_q), "streams": len(by_s),
        "ms_by_queue": {str(k): round(v / 1e6, 1) for k, v in sorted(by_q.items(), key=lambda kv: -kv[1])[:20]},
        "top_kernels_ms": {k: [v[0], round(v[1] / 1e6, 1)] for k, v in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:14]},
    }


if __name__ == "__main__":
    print(json.dumps(analyse(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0, sys.argv[3] if len(sys.argv) > 3 else None)))
