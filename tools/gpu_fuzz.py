"""Randomised differential test on the GPU: random data shapes, sizes, levels and speculation settings
(segment size, window, hand-off group, poll budget) -- every stream must equal the oracle's byte for byte.
Speculation settings change timing and interleaving only; any difference is a bug."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data  # noqa: E402
import _oracle  # noqa: E402
import orz_amd  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    encs = {lv: orz_amd.StreamEncoder(0, lv, mode="exact") for lv in (0, 1, 2)}
    makers = [("text", _data.text), ("mixed", _data.mixed), ("zeros", lambda n, seed=0: _data.zeros_noise(n)),
              ("random", _data.random_bytes), ("p1", lambda n, seed=0: _data.periodic(n, 1)),
              ("p4", lambda n, seed=0: _data.periodic(n, 4)), ("p7", lambda n, seed=0: _data.periodic(n, 7))]
    t0 = time.time()
    cases = bad = 0
    while time.time() - t0 < budget:
        name, mk = makers[int(rng.integers(0, len(makers)))]
        n = int(10 ** rng.uniform(3.5, 6.7))
        lv = int(rng.integers(0, 3))
        seg = int(rng.choice([8, 13, 31, 47, 62]))
        win = int(rng.choice([1, 7, 100, 1024, 3072, 4096]))
        chain = int(rng.choice([1, 2, 8, 48, 64]))
        polls = int(rng.choice([0, 3, 250]))
        data = mk(n, seed=int(rng.integers(0, 1 << 30)))
        os.environ["ORZ_CHAIN"] = str(chain)
        os.environ["ORZ_POLLS"] = str(polls)
        enc = encs[lv]
        enc.set_tuning(seg, win)
        out = enc.encode(data)
        ok = out == _oracle.encode(data, lv)
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", name, n, "level", lv, "seg", seg, "win", win, "chain", chain, "polls", polls, flush=True)
    print("fuzz: %d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
