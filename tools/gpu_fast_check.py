"""GPU check of the fast parse mode: round trip through the oracle decoder, size against the oracle encoder
(the +-0.5 % band of BASELINE.json), timing.  Usage: python tools/gpu_fast_check.py [nbytes] [level] [tile] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _data  # noqa: E402
import _oracle  # noqa: E402
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def main():
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    enc = orz_amd.StreamEncoder(device=0, level=level, mode="fast", tile_bytes=tile, rounds=rounds)
    print(json.dumps(enc.config()))
    cases = {"text30k": _data.text(30_000), "mixed200k": _data.mixed(200_000, seed=3), "zeros1M": _data.zeros_noise(1_000_000),
             "random100k": _data.random_bytes(100_000), "empty": b"", "one": b"a"}
    cases["enwik_like"] = corpus.enwik_like(nbytes)
    for name, d in cases.items():
        t = time.time()
        out, st = enc.encode(d, stats=True)
        dt = time.time() - t
        ref = _oracle.encode(d, level) if len(d) <= 40_000_000 else None
        back, _ = _oracle.decode(out)
        rec = {"case": name, "n": len(d), "fast": len(out), "oracle": len(ref) if ref else None,
               "delta_pct": round(100.0 * (len(out) - len(ref)) / max(1, len(ref)), 3) if ref else None,
               "roundtrip": back == d, "wall_s": round(dt, 3), "device_ms": round(st["total_ms"], 2),
               "MBps": round(len(d) / 1e6 / max(1e-9, st["total_ms"] / 1e3), 1), "steps": st["sweeps"], "repairs": st["seg_evals"],
               "prep_s": round(st["t_prep_s"], 4), "parse_s": round(st["t_parse_s"], 4), "post_s": round(st["t_post_s"], 4)}
        print(json.dumps(rec), flush=True)
        assert back == d, name


if __name__ == "__main__":
    main()
