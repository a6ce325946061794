"""Sanitizer run of the encoder's kernel bodies (parse, rank, tail stage) on the host emulation backend (TEST INFRASTRUCTURE; see run.sh)."""
import ctypes
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _data  # noqa: E402
import _oracle  # noqa: E402

lib = ctypes.CDLL(os.environ.get("ORZ_EMU_ASAN", "/tmp/libemu_asan.so"))


def enc(data, cfg=(15,9,6), seg=62, win=128, order=2):
    dst = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(); st = (ctypes.c_ulonglong*5)()
    rc = lib.emu_encode(bytes(data), ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], seg, win, order, ctypes.byref(dst), ctypes.byref(n), st)
    assert rc == 0
    out = ctypes.string_at(dst, n.value); lib.emu_free(dst); return out
for name, data, lv, seg, win in [("mixed", _data.mixed(30_000, seed=2), 1, 62, 128), ("zeros", _data.zeros_noise(20_000), 2, 62, 64), ("p3", _data.periodic(6_000, 3), 0, 17, 200), ("tiny", b"abc", 1, 62, 64), ("one", b"x", 2, 8, 300)]:
    cfg = _oracle.LEVELS[lv]
    out = enc(data, cfg, seg, win)
    print(name, len(out), out == _oracle.encode(data, lv))


# fast parse mode (orz_fast.h): rows / rounds / path / repair kernels under the sanitizers; parity = oracle decode
def enc_fast(data, cfg=(15, 9, 6), tile=0, rounds=0):
    dst = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(); st = (ctypes.c_ulonglong*5)()
    rc = lib.emu_encode_fast(bytes(data), ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], tile, rounds, ctypes.byref(dst), ctypes.byref(n), st)
    assert rc == 0
    out = ctypes.string_at(dst, n.value); lib.emu_free(dst); return out
for name, data, lv, tile, rounds in [("mixed", _data.mixed(60_000, seed=2), 1, 4096, 3), ("zeros", _data.zeros_noise(50_000), 2, 8192, 4),
                                     ("p3", _data.periodic(9_000, 3), 0, 4096, 2), ("text", _data.text(70_000, seed=4), 1, 0, 0),
                                     ("tiny", b"abc", 1, 0, 0), ("one", b"x", 2, 0, 0), ("empty", b"", 1, 0, 0),
                                     ("odd", _data.mixed(4097 + 63, seed=8), 1, 4096, 4)]:
    out = enc_fast(data, _oracle.LEVELS[lv], tile, rounds)
    print("fast", name, len(out), _oracle.decode(out)[0] == data)

# two blocks + a short tail with the default schedule (history lists, window slide, retired tiles, graph-free launch order):
# slow under the sanitizers (minutes); ORZ_SAN_BIG=0 skips it
if os.environ.get("ORZ_SAN_BIG", "1") != "0":
    import corpus  # noqa: E402

    data = corpus.enwik_like(36_000_000)
    out = enc_fast(data, _oracle.LEVELS[1])
    print("fast two blocks + tail", len(out), _oracle.decode(out)[0] == data)
