#!/usr/bin/env python3
"""DESIGN.md section 6's kernel table from a bench line (bench.py's `kernel_table`: HIP events around EVERY launch of one profiled
pass, orz_stream_get_kernel_table) -- optionally beside a rocprofv3 kernel-stats CSV of the same workload (tools/rocpd_summary.py),
whose launch times carry no event overhead, and a PMC traffic file (tools/pmc_summary.py).
    python tools/kernel_table.py BENCH.json [--stats KERNEL_STATS.csv --blocks 24] [--pmc PMC.json] [--min-ms 0.05]
Prints a markdown table: kernel | launches per 16 MiB block | average launch | ms per block | HBM bytes per block | GB/s."""
import argparse
import csv
import json
import re


def short(name):
    m = re.search(r"orz_(?:thread_kernel(?:_occ)?|wave_kernel|group_kernel)INS_\d+([A-Za-z0-9]+?)(?:I[a-z]E)?EEEvT_", name)
    if m:
        return m.group(1)
    if "symrank" in name:
        return "orz_symrank_kernel"
    if "radix_sort" in name:
        return "(radix sort)"
    if "scan" in name:
        return "(scan)"
    if "fillBuffer" in name:
        return "(fill)"
    if "copyBuffer" in name:
        return "(copy)"
    return name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench")
    ap.add_argument("--stats")
    ap.add_argument("--blocks", type=float, default=24.0, help="16 MiB blocks the stats file covers (bench --steps 2 --warmup 1 + profile pass = 24)")
    ap.add_argument("--pmc")
    ap.add_argument("--min-ms", type=float, default=0.05)
    ap.add_argument("--into", help="a markdown file: the table replaces what stands between <!-- KERNEL_TABLE_BEGIN --> and <!-- KERNEL_TABLE_END -->")
    a = ap.parse_args()
    d = json.loads(open(a.bench).read().strip().splitlines()[-1])
    rows = {r["kernel"]: r for r in d["kernel_table"]["rows"]}
    prof = {}
    if a.stats:
        for r in csv.DictReader(open(a.stats)):
            k = short(r["name"])
            c, t = prof.get(k, (0, 0.0))
            prof[k] = (c + int(r["calls"]), t + float(r["total_ns"]))
    pmc = json.load(open(a.pmc)).get("by_name", {}) if a.pmc else {}
    pmcs = {}
    for k, v in pmc.items():
        m = re.search(r"<([A-Za-z0-9]+)>", k)
        pmcs.setdefault(m.group(1) if m else short(k), v)
    out = []
    _print = out.append
    _print("| kernel | launches / block | avg launch (events) | ms / block (events) | ms / block (rocprofv3) | HBM bytes / launch (PMC) | GB/s |")
    _print("|---|---|---|---|---|---|---|")
    tot_e = tot_p = 0.0
    lib = {"(fill)": "(fill)", "(scan)": "(scan)", "(scan, running maximum)": "(scan)", "(radix sort, pairs)": "(radix sort)",
           "(radix sort, 64-bit keys)": "(radix sort)", "(radix sort, items by context)": "(radix sort)", "orz_symrank_kernel (+ guard)": "orz_symrank_kernel"}
    shown = set()
    all_prof = sum(t for k, (c, t) in prof.items() if "symrank" not in k) / 1e6 / a.blocks
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_block"]):
        pk = lib.get(k, k.split("<")[0])
        p = prof.get(pk) if pk not in shown else None  # (the library calls of one kind share a row of the stats file: shown once)
        shown.add(pk)
        pms = p[1] / 1e6 / a.blocks if p else None
        if "symrank" not in k:
            tot_e += r["ms_per_block"]
            pass
        if r["ms_per_block"] < a.min_ms:
            continue
        tr = pmcs.get(k.split("<")[0]) or pmcs.get(pk) or r.get("hbm_bytes_per_launch")
        gbs = tr / (r["avg_launch_us"] * 1e-6) / 1e9 if tr else None
        _print("| `%s` | %.1f | %.1f us | %.3f | %s | %s | %s |" % (k, r["launches_per_block"], r["avg_launch_us"], r["ms_per_block"],
                                                              "%.3f" % pms if pms is not None else "", "%.1f MB" % (tr / 1e6) if tr else "", "%.0f" % gbs if gbs else ""))
    _print("| **sum without the symbol ranking** | | | **%.2f** | **%s** | | |" % (tot_e, "%.2f" % all_prof if prof else ""))
    text = "\n".join(out) + "\n"
    if a.into:
        doc = open(a.into).read()
        b0, b1 = "<!-- KERNEL_TABLE_BEGIN -->", "<!-- KERNEL_TABLE_END -->"
        i, j = doc.index(b0) + len(b0), doc.index(b1)
        open(a.into, "w").write(doc[:i] + "\n" + text + doc[j:])
    else:
        print(text, end="")


if __name__ == "__main__":
    main()
