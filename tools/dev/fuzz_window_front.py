"""dev: two-block inputs on the HOST EMULATION whose second block refers to the oldest history positions (the first bytes of
the stream, which sit at window offsets 0..7 after the slide) from rare contexts -- the neighbourhood of the defect fixed in
round 3 (the context of window offset 1 needs the byte before the window).  Every stream is decoded by the
oracle.  The filler is one repeated byte, so a 16 MiB block costs the emulation half a minute.  (Power: limited -- on the
code WITHOUT the fix the same 16 cases also decode; the repairs cut the near cases this generator makes.  The reproducer of
the defect is a real member: tests/test_gpu_fast.py, tools/dev/soak_members.py.)
    python tools/dev/fuzz_window_front.py [cases=12] [seed=1]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import _oracle  # noqa: E402
import emu_size  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = emu_size.emu_lib()
PUNCT = b"-+=_;:!?#@/"
ALNUM = b"aZ09qM"
OTHER = b" \n\t.,()"
bad = 0
for c in range(cases):
    keys = [bytes(rng.choice(b"QWERTYUIOPASDFGHJKLZXCVBNM") for _ in range(rng.randint(6, 40))) for _ in range(rng.randint(1, 4))]
    head = b""
    for k in keys:  # the stream starts with short items in rare contexts: <alnum or not><punctuation><key>
        head += bytes([rng.choice(ALNUM + OTHER), rng.choice(PUNCT)]) + k
    head = head[rng.randint(0, 2):]  # the first item starts move by a byte or two
    fill = bytes([rng.choice(b"xy.")])
    second = fill * rng.randint(100, 3000)
    for _ in range(rng.randint(1, 6)):
        k = rng.choice(keys)
        for i in range(rng.randint(0, 90)):  # decoys: same context and hash entry, texts part after four bytes
            second += bytes([rng.choice(OTHER + ALNUM), rng.choice(PUNCT)]) + k[:4] + bytes([48 + i % 10, 65 + i % 26]) + fill * rng.randint(1, 60)
        second += bytes([rng.choice(OTHER + ALNUM), rng.choice(PUNCT)]) + k + fill * rng.randint(1, 500)
    data = head + fill * ((1 << 24) - len(head)) + second
    t0 = time.time()
    out, st = emu_size.fast(lib, data, (15, 9, 6))
    try:
        back, used = _oracle.decode(out)
        ok = back == data and used == len(out)
    except Exception as e:  # noqa: BLE001
        ok = False
        print("   ", e)
    bad += 0 if ok else 1
    print("case %d: %d bytes -> %d, repairs %d, %s, %.0f s" % (c, len(data), len(out), st[2], "ok" if ok else "INVALID", time.time() - t0), flush=True)
    if not ok:
        open("/tmp/fuzz_front_case%d.bin" % c, "wb").write(data)
print("%d cases, %d invalid" % (cases, bad))
