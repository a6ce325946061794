"""BASELINE.json configs[2] and configs[4] on one GPU, through the product path only:
  C2  1 GB of enwik8-shaped text (the 100 MB bench workload repeated: no cross-copy matches, the window is 32 MiB),
      -l2, independent members (64 MiB by default: the cold start of a member costs ~0.3 % of the size there)
  C4  1 GB of zeros with 1 % uniform noise (splitmix64, seed 0x6f727a), -l2 -- degenerate regime of the
      symbol-ranking / Huffman path
Reports encode MB/s, ratio, and the size-independent parity property available without the oracle: the
container decodes (library host decoder, member by member) to the input, bit for bit.  The byte-for-byte
comparison with the oracle at full size lives in tests/ (smaller sizes) -- tools never touch oracle/.

  python tools/gpu_configs.py [total_bytes=1000000000] [jobs=8] [member_bytes=67108864] [level=2]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def run(name, data, level, jobs, member=1 << 26):
    enc = orz_amd.MemberEncoder(device=0, level=level, jobs=jobs)
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)  # warm-up: allocations, first launches
    t0 = time.time()
    blob, n = enc.encode(data, member_bytes=member)
    t_enc = time.time() - t0
    enc.close()
    t0 = time.time()
    back, m = orz_amd.decode_members(blob)
    t_dec = time.time() - t0
    ok = m == n and hashlib.sha256(back).digest() == hashlib.sha256(data).digest()
    print(json.dumps({"config": name, "bytes": len(data), "level": level, "members": n, "member_bytes": member, "jobs": jobs,
                      "compressed": len(blob), "ratio": round(len(blob) / len(data), 5),
                      "encode_MBps": round(len(data) / t_enc / 1e6, 1), "encode_s": round(t_enc, 2),
                      "host_decode_MBps": round(len(data) / t_dec / 1e6, 1), "round_trip_exact": bool(ok)}), flush=True)
    return ok


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    member = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
    level = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    text = corpus.enwik_like(total)
    ok = run("C2: enwik8-shaped text, %d bytes, -l%d, members of %d bytes, %d encoders on one GPU" % (total, level, member, jobs), text, level, jobs, member)
    del text
    ok &= run("C4: zeros + 1 pct noise, %d bytes, -l%d, members of %d bytes, %d encoders on one GPU" % (total, level, member, jobs), corpus.zeros_noise(total), level, jobs, member)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
