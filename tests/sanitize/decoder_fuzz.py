"""Sanitizer run of the device decoder's kernel body: corrupted containers must be rejected or decoded, never read or write out of bounds on the host emulation backend (TEST INFRASTRUCTURE; see run.sh)."""
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


def dec(blob, slots=4):
    dst = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t(); m = ctypes.c_size_t(); err = ctypes.create_string_buffer(256)
    rc = lib.emu_decode_members(blob, ctypes.c_size_t(len(blob)), slots, ctypes.byref(dst), ctypes.byref(n), ctypes.byref(m), err, ctypes.c_size_t(256))
    if rc: return None
    out = ctypes.string_at(dst, n.value); lib.emu_free(dst); return out
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
parts = [_data.mixed(30_000, seed=1), _data.zeros_noise(20_000), _data.random_bytes(5_000), _data.periodic(9_000, 3)]
good = b"".join(_oracle.encode(p, i % 3) for i, p in enumerate(parts))
assert dec(good) == b"".join(parts)
ok = bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    b = bytearray(good)
    k = rnd.randrange(4)
    if k == 0:
        for _ in range(rnd.randrange(1, 6)): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif k == 1:
        at = rnd.randrange(len(b)); b[at:at + rnd.randrange(1, 50)] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 50)))
    elif k == 2:
        b = b[:rnd.randrange(len(b))]
    else:
        at = rnd.randrange(len(b)); del b[at:at + rnd.randrange(1, 2000)]
    r = dec(bytes(b))
    if r is None: bad += 1
    else: ok += 1
print("corrupted containers:", ok, "decoded to something,", bad, "rejected; no crash")
