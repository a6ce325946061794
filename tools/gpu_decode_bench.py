"""Symmetric round trip on one GPU: the bench.py text workload is encoded as members of a
given size (GPU members encoder), then the container is decoded (a) by the device decoder -- one member per
wavefront -- and (b) by the library's host decoder, one thread.  Prints one JSON line per member size.

  python tools/gpu_decode_bench.py [total_bytes] [member_bytes ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    sizes = [int(a) for a in sys.argv[2:]] or [65536, 262144, 1 << 20]
    data = corpus.text_corpus(total)
    for mb in sizes:
        enc = orz_amd.MemberEncoder(device=0, level=1, jobs=4)
        t0 = time.time()
        blob, n = enc.encode(data, member_bytes=mb)
        t_enc = time.time() - t0
        enc.close()
        orz_amd.decode_members_device(blob[: len(blob) // 50 or len(blob)] if False else blob[:0])  # (context warm-up)
        t0 = time.time()
        out, m, st = orz_amd.decode_members_device(blob, stats=True)
        t_dev = time.time() - t0
        ok = out == data and m == n
        t0 = time.time()
        host, _ = orz_amd.decode_members(blob)
        t_host = time.time() - t0
        print(json.dumps({
            "bytes": len(data), "member_bytes": mb, "members": n, "compressed": len(blob), "ratio": round(len(blob) / len(data), 4),
            "encode_s": round(t_enc, 3),
            "device_decode_kernel_MBps": round(len(data) / (st["kernel_ms"] / 1e3) / 1e6, 1) if st["kernel_ms"] else None,
            "device_decode_kernel_ms": round(st["kernel_ms"], 2), "device_decode_wall_s": round(t_dev, 3),
            "device_decode_launches": st["launches"],
            "host_decode_MBps_1_thread": round(len(data) / t_host / 1e6, 1), "exact": bool(ok and host == data)}), flush=True)


if __name__ == "__main__":
    main()
