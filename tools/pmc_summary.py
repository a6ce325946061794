"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately,
as MI355X_MICROARCH.md prescribes: TCC slots do not fit both).  Values are KiB per dispatch as rocprofv3
reports them; FETCH_SIZE under-counts wide coalesced 16 B/lane streams by 2x on gfx950 (guide, section HBM) --
the parse kernel's accesses are narrow and scattered; tools/pmc_calibrate.py measures what the counter
reports for that pattern on the box (128 B per scattered 32-byte row, 1.00 x for a streaming read), so the
figure is the counter as is: line fetches x 128 B."""
import json
import sqlite3
import sys


def per_kernel(db_path):
    cur = sqlite3.connect(db_path).cursor()
    q = """select k.kernel_name, count(*), sum(e.value) from rocpd_pmc_event e
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name"""
    return {r[0]: (r[1], r[2]) for r in cur.execute(q)}


if __name__ == "__main__":
    fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = per_kernel(fetch), per_kernel(write)
    rows = []
    for name in sorted(set(f) | set(w), key=lambda n: -(f.get(n, (0, 0))[1] + w.get(n, (0, 0))[1])):
        nf, sf = f.get(name, (0, 0.0))
        nw, sw = w.get(name, (0, 0.0))
        rows.append({"kernel": name[:100], "dispatches": int(max(nf, nw)),
                     "fetch_KiB_per_dispatch": round(sf / nf, 1) if nf else None,
                     "write_KiB_per_dispatch": round(sw / nw, 1) if nw else None})
    res = {"units": "KiB per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes; scattered loads count 128 B per line, see pmc_calibrate.py)", "kernels": rows[:80]}
    res["total_bytes_all_kernels"] = int(sum(((r["fetch_KiB_per_dispatch"] or 0) + (r["write_KiB_per_dispatch"] or 0)) * 1024 * r["dispatches"] for r in rows))
    # HBM bytes per launch under the names bench.py uses for its roofline rows
    short = {"FastEval": "orz_thread_kernel<FastEval>", "ParseWave": "orz_wave_kernel<ParseWave>", "orz_symrank_kernel": "orz_symrank_kernel",
             "FastRowsWave": "orz_wave_kernel<FastRowsWave>", "PathUpWave": "orz_wave_kernel<PathUpWave>"}
    res["by_name"] = {}
    import re
    for r in rows:  # every kernel of the library under the functor's name, as bench.py's kernel_table spells it (round 5)
        m = re.search(r"orz_(?:thread_kernel(?:_occ)?|wave_kernel|group_kernel)INS_\d+([A-Za-z0-9]+?)(?:I[a-z]E)?EEEvT_", r["kernel"])
        if m:
            res["by_name"]["<%s>" % m.group(1)] = int(((r["fetch_KiB_per_dispatch"] or 0) + (r["write_KiB_per_dispatch"] or 0)) * 1024)
    for r in rows:
        for key, name in short.items():
            if key in r["kernel"]:
                res["by_name"][name] = int(((r["fetch_KiB_per_dispatch"] or 0) + (r["write_KiB_per_dispatch"] or 0)) * 1024)
        if "ParseWave" in r["kernel"]:
            res["parse_wave_hbm_bytes_per_launch"] = int(((r["fetch_KiB_per_dispatch"] or 0) + (r["write_KiB_per_dispatch"] or 0)) * 1024)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])
