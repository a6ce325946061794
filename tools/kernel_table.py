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
    a = ap.parse_args()
    d = json.loads(open(a.bench).read().strip().splitlines()[-1])
    rows = {r["kernel"].split(" ")[0].split("<")[0]: r for r in d["kernel_table"]["rows"]}
    prof = {}
    if a.stats:
        for r in csv.DictReader(open(a.stats)):
            k = short(r["name"])
            c, t = prof.get(k, (0, 0.0))
            prof[k] = (c + int(r["calls"]), t + float(r["total_ns"]))
    pmc = json.load(open(a.pmc)).get("by_name", {}) if a.pmc else {}
    pmcs = {short(k) if "orz_" in k and "<" not in k else re.sub(r".*<(.*)>", r"\\1", k): v for k, v in pmc.items()}
    print("| kernel | launches / block | avg launch (events) | ms / block (events) | ms / block (rocprofv3) | HBM bytes / launch (PMC) | GB/s |")
    print("|---|---|---|---|---|---|---|")
    tot_e = tot_p = 0.0
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_block"]):
        p = prof.get(k)
        pms = p[1] / 1e6 / a.blocks if p else None
        if "symrank" not in k:
            tot_e += r["ms_per_block"]
            tot_p += pms or 0.0
        if r["ms_per_block"] < a.min_ms:
            continue
        tr = pmcs.get(k) or r.get("hbm_bytes_per_launch")
        gbs = tr / (r["avg_launch_us"] * 1e-6) / 1e9 if tr else None
        print("| `%s` | %.1f | %.1f us | %.3f | %s | %s | %s |" % (k, r["launches_per_block"], r["avg_launch_us"], r["ms_per_block"],
                                                              "%.3f" % pms if pms is not None else "", "%.1f MB" % (tr / 1e6) if tr else "", "%.0f" % gbs if gbs else ""))
    print("| **sum without the symbol ranking** | | | **%.2f** | **%s** | | |" % (tot_e, "%.2f" % tot_p if tot_p else ""))


if __name__ == "__main__":
    main()
