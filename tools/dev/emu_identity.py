"""dev: SHA-256 of the host emulation's fast-mode streams for a fixed set of inputs -- `save` writes the table, `check` compares.
A refactor of the parse that must not change a byte (lists instead of grids, fused kernels) is checked with this in a minute."""
import ctypes, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import _data, corpus

def lib():
    so = os.path.join(ROOT, "build", "libemu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_backend.cpp", "simt.h")]
    srcs += [os.path.join(ROOT, "orz_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "orz_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    return ctypes.CDLL(so)

CASES = {
    "text3M_l1": (lambda: corpus.enwik_like(3_000_000), (15, 9, 6), 0),
    "text2M_l0_t64k": (lambda: corpus.enwik_like(2_000_000), (5, 3, 2), 65536),
    "mixed2M_l1": (lambda: _data.mixed(2_000_000), (15, 9, 6), 0),
    "mixed1M_l0_t16k": (lambda: _data.mixed(1_000_000, seed=4), (5, 3, 2), 16384),
    "zeros2M_l2": (lambda: _data.zeros_noise(2_000_000), (45, 27, 18), 0),
    "period7_500k_l1": (lambda: _data.periodic(500_000, 7), (15, 9, 6), 0),
    "random300k_l1": (lambda: _data.random_bytes(300_000), (15, 9, 6), 0),
    "synth1M_l2": (lambda: _data.text(1_000_000), (45, 27, 18), 0),
}

def run(L, data, cfg, tile):
    dst = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(); st = (ctypes.c_ulonglong * 5)()
    rc = L.emu_encode_fast(data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], tile, 0, ctypes.byref(dst), ctypes.byref(n), st)
    assert rc == 0
    out = ctypes.string_at(dst, n.value); L.emu_free(dst)
    return out, list(st)

def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    path = os.path.join(ROOT, "build", "emu_identity.json")
    L = lib()
    want = json.load(open(path)) if mode == "check" else {}
    got, bad = {}, 0
    for name, (mk, cfg, tile) in CASES.items():
        if len(sys.argv) > 2 and name not in sys.argv[2:]: continue
        t0 = time.time(); out, st = run(L, mk(), cfg, tile)
        got[name] = [hashlib.sha256(out).hexdigest(), len(out), st[2]]
        flag = "" if mode != "check" else ("ok" if want.get(name) == got[name] else "DIFFERS from %r" % (want.get(name),))
        bad += flag.startswith("DIFF")
        print("%-20s %9d bytes  repairs %6d  %.1fs %s" % (name, len(out), st[2], time.time() - t0, flag), flush=True)
    if mode == "save": json.dump(got, open(path, "w"), indent=1)
    sys.exit(1 if bad else 0)
main()
