#!/usr/bin/env python3
"""benchmark_tool.py -- the reference's `benchmark-tool` for this repo.

Follows /root/reference/benchmark-tool/src/main.rs:22-121: every compressor is run as a CHILD PROCESS with
the bench file on stdin and a temp file on stdout, three rounds of encode + decode, the decoded file's MD5
is compared with the input's (:103-108), the row keeps the MINIMUM time of the rounds (:111-112) and the
table is sorted by compressed size (:54) and printed as markdown (:55).

Differences, on purpose:
  * the reference reports the children's user CPU time (getrusage(RUSAGE_CHILDREN), :116-121).  For a GPU
    encoder that number is the host's share only, so the table carries BOTH user time and wall time and
    MB/s is computed from wall time;
  * rows: `orz -l0/-l1/-l2` = this repo's `bin/orz` (HIP encoder, host decoder), `oracle -lN` = the CPU
    restatement of the reference (oracle/, test infrastructure, timed here as the CPU baseline only), then
    whichever of gzip / bzip2 / xz / zstd / brotli exist on the box (the reference's list, :23-36).

  python tools/benchmark_tool.py <bench-file> [--rounds 3] [--skip-oracle] [--json out.json]
  python tools/benchmark_tool.py --corpus 100000000        # the bench.py workload written to a temp file
"""
import argparse
import hashlib
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORZ = os.path.join(ROOT, "bin", "orz")
ORACLE = os.path.join(ROOT, "oracle", "orz_oracle")


def children_utime():
    return resource.getrusage(resource.RUSAGE_CHILDREN).ru_utime


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run(cmd, src, dst, files_as_args):
    """one child; returns (user seconds, wall seconds)"""
    u0, t0 = children_utime(), time.time()
    if files_as_args:  # the oracle CLI takes file names
        rc = subprocess.call(cmd + [src, dst], stderr=subprocess.DEVNULL)
    else:
        with open(src, "rb") as fi, open(dst, "wb") as fo:
            rc = subprocess.call(cmd, stdin=fi, stdout=fo, stderr=subprocess.DEVNULL)
    if rc != 0:
        raise RuntimeError("%s: exit status %d" % (" ".join(cmd), rc))
    return children_utime() - u0, time.time() - t0


def bench(tmp, path, name, enc, dec, rounds, want_md5, files_as_args=False):
    sys.stderr.write("start benchmarking %s...\n" % name)
    eo, do = os.path.join(tmp, "enc_output"), os.path.join(tmp, "dec_output")
    enc_t, dec_t = [], []
    for i in range(rounds):
        enc_t.append(run(enc, path, eo, files_as_args))
        sys.stderr.write(" => round %d: finished encoding: user=%.3fs wall=%.3fs\n" % (i, *enc_t[-1]))
        dec_t.append(run(dec, eo, do, files_as_args))
        sys.stderr.write(" => round %d: finished decoding: user=%.3fs wall=%.3fs\n" % (i, *dec_t[-1]))
        if md5(do) != want_md5:
            raise RuntimeError("%s.decode: wrong result" % name)
    size = os.path.getsize(eo)
    return {"name": name, "size": size,
            "enc_user": min(t[0] for t in enc_t), "enc_wall": min(t[1] for t in enc_t),
            "dec_user": min(t[0] for t in dec_t), "dec_wall": min(t[1] for t in dec_t)}


def table(rows, nbytes):
    out = ["| name | compressed size | encode user | encode wall | encode MB/s | decode user | decode wall | decode MB/s |",
           "|------|-----------------|-------------|-------------|-------------|-------------|-------------|-------------|"]
    for r in sorted(rows, key=lambda r: r["size"]):
        out.append("| %s | %s | %.3fs | %.3fs | %.1f | %.3fs | %.3fs | %.1f |" % (
            r["name"], format(r["size"], ","), r["enc_user"], r["enc_wall"], nbytes / r["enc_wall"] / 1e6,
            r["dec_user"], r["dec_wall"], nbytes / r["dec_wall"] / 1e6))
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_file", nargs="?")
    ap.add_argument("--corpus", type=int, default=0, help="use the first N bytes of the bench.py text workload")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--skip-orz", action="store_true", help="no GPU rows (a box without a HIP device)")
    ap.add_argument("--skip-others", action="store_true", help="orz rows only")
    ap.add_argument("--other-rounds", type=int, default=0, help="rounds for gzip/bzip2/xz/... (default: --rounds)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--exact-rows", action="store_true", help="also time `orz --mode exact` (one round)")
    args = ap.parse_args()
    if not args.bench_file and not args.corpus:
        ap.error("usage: benchmark_tool.py <bench-file> | --corpus N")

    with tempfile.TemporaryDirectory() as tmp:
        path = args.bench_file
        if args.corpus:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import corpus

            path = os.path.join(tmp, "bench_input")
            with open(path, "wb") as f:
                f.write(corpus.enwik_like(args.corpus))
        nbytes = os.path.getsize(path)
        want = md5(path)
        rows = []
        for lv in (() if args.skip_orz else (0, 1, 2)):  # the library's default mode (fast unless ORZ_MODE=exact), then the exact one
            rows.append(bench(tmp, path, "**orz -l%d** (MI355X)" % lv, [ORZ, "encode", "-s", "-l%d" % lv], [ORZ, "decode", "-s"],
                              args.rounds, want))
        for lv in (() if (args.skip_orz or not args.exact_rows) else (0, 1, 2)):
            rows.append(bench(tmp, path, "orz -l%d --mode exact (MI355X, the reference's parse)" % lv, [ORZ, "encode", "-s", "--mode", "exact", "-l%d" % lv],
                              [ORZ, "decode", "-s"], 1, want))
        if not args.skip_oracle and os.path.exists(ORACLE):
            for lv in (0, 1, 2):
                rows.append(bench(tmp, path, "oracle -l%d (CPU restatement, 1 thread)" % lv, [ORACLE, "encode", "-l%d" % lv],
                                  [ORACLE, "decode"], args.rounds, want, files_as_args=True))
        if not args.skip_others:
            others = [("gzip -6", ["gzip", "-6"], ["gzip", "-d"]), ("bzip2 -9", ["bzip2", "-9"], ["bzip2", "-d"]),
                      ("xz -6", ["xz", "-6"], ["xz", "-d"]), ("zstd -10", ["zstd", "-10"], ["zstd", "-d"]),
                      ("zstd -19", ["zstd", "-19"], ["zstd", "-d"]), ("brotli -9", ["brotli", "-9"], ["brotli", "-d"])]
            for name, enc, dec in others:
                if shutil.which(enc[0]):
                    rows.append(bench(tmp, path, name, enc, dec, args.other_rounds or args.rounds, want))
        print("bench file: %d bytes, md5 %s, %d rounds, minimum taken; MB/s from wall time\n" % (nbytes, want, args.rounds))
        print(table(rows, nbytes))
        if args.json:
            with open(args.json, "w") as f:
                json.dump({"bytes": nbytes, "md5": want, "rounds": args.rounds, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
