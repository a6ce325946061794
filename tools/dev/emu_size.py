"""Size of the fast parse on the HOST EMULATION (tests/emu: the very kernel bodies of orz_amd/csrc as CPU loops) against the
oracle's encoder -- the place to try schedule / search changes without GPU time.
    python tools/dev/emu_size.py [--mb 4] [--shape text|mixed|zeros|source] [--level 1] [--tile 131072] [--rounds 4]
Environment knobs of the parse (ORZ_FAST_*) apply as on the GPU."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LEVELS = {0: (5, 3, 2), 1: (15, 9, 6), 2: (45, 27, 18)}


def emu_lib():
    so = os.path.join(ROOT, "build", "libemu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_backend.cpp", "simt.h")]
    srcs += [os.path.join(ROOT, "orz_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "orz_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    return ctypes.CDLL(so)


def fast(lib, data, cfg, tile=0, rounds=0):
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    st = (ctypes.c_ulonglong * 5)()
    rc = lib.emu_encode_fast(data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], tile, rounds, ctypes.byref(dst), ctypes.byref(n), st)
    assert rc == 0
    out = ctypes.string_at(dst, n.value)
    lib.emu_free(dst)
    return out, list(st)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=4.0)
    ap.add_argument("--shape", default="text")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=0)
    ap.add_argument("--no-decode", action="store_true")
    a = ap.parse_args()
    import _data
    import _oracle
    import corpus

    n = int(a.mb * 1e6)
    if a.shape == "text":
        data = corpus.enwik_like(n)
    elif a.shape == "mixed":
        data = _data.mixed(n, seed=17)
    elif a.shape == "zeros":
        data = _data.zeros_noise(n)
    elif a.shape == "source":
        data = corpus.source_like(n) if hasattr(corpus, "source_like") else corpus.build(n)
    else:
        raise SystemExit("shape?")
    data = bytes(data)
    lib = emu_lib()
    t0 = time.time()
    out, st = fast(lib, data, LEVELS[a.level], a.tile, a.rounds)
    t1 = time.time()
    ref = len(_oracle.encode(data, a.level))
    ok = None
    if not a.no_decode:
        back, used = _oracle.decode(out)
        ok = back == data and used == len(out)
    print("%s %.1f MB -l%d: fast %d B, oracle %d B, delta %+.4f %%, repairs %d, items %d, steps %d, round trip %s, emu %.1f s"
          % (a.shape, a.mb, a.level, len(out), ref, 100.0 * (len(out) - ref) / ref, st[2], st[3], st[1], ok, t1 - t0))


if __name__ == "__main__":
    main()
