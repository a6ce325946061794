"""Randomised test of the FAST parse mode on the GPU: random data shapes, sizes, levels, tile sizes, rounds, row widths
and far-search reach.  Every stream must (1) decode with the oracle's decoder to the input, (2) be reproduced byte for
byte by the oracle's plan-driven encoder fed the GPU's parse (every item representable, post stage exact); the size
relative to the oracle encoder is reported (worst case per shape)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data  # noqa: E402
import _oracle  # noqa: E402
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    makers = [("text", _data.text), ("mixed", _data.mixed), ("zeros", lambda n, seed=0: _data.zeros_noise(n)),
              ("random", _data.random_bytes), ("p1", lambda n, seed=0: _data.periodic(n, 1)),
              ("p4", lambda n, seed=0: _data.periodic(n, 4)), ("p7", lambda n, seed=0: _data.periodic(n, 7)),
              ("enwik", lambda n, seed=0: corpus.enwik_like(4_000_000)[seed % 1_000_000:][:n])]
    t0 = time.time()
    cases = bad = 0
    worst = {}
    while time.time() - t0 < budget:
        name, mk = makers[int(rng.integers(0, len(makers)))]
        n = int(10 ** rng.uniform(0.0, 6.4))
        lv = int(rng.integers(0, 3))
        tile = int(rng.choice([4096, 8192, 65536, 131072]))
        rounds = int(rng.choice([1, 2, 3, 4, 6]))
        os.environ["ORZ_FAST_K"] = str(int(rng.choice([64, 128, 192])))
        os.environ["ORZ_FAST_FAR"] = str(int(rng.choice([0, 256, 4096])))
        os.environ["ORZ_GRAPHS"] = str(int(rng.integers(0, 2)))
        data = mk(n, seed=int(rng.integers(0, 1 << 30)))
        enc = orz_amd.StreamEncoder(0, lv, mode="fast", tile_bytes=tile, rounds=rounds)
        enc.set_item_trace(True)
        try:
            out = enc.encode(data)
            tr = enc.item_trace()
        finally:
            enc.close()
        ok = True
        try:
            ok = _oracle.decode(out)[0] == data
            if ok and n <= 1_500_000:
                ok = _oracle.encode_plan(data, _oracle.plan_from_trace(tr, len(data))) == out
        except Exception as e:  # noqa: BLE001
            ok = False
            print("EXC", e)
        cases += 1
        ref = len(_oracle.encode(data, lv))
        d = 100.0 * (len(out) - ref) / max(ref, 1)
        if n >= 100_000:
            worst[name] = max(worst.get(name, -100.0), d)
        if not ok:
            bad += 1
            print("FAIL", name, n, "level", lv, "tile", tile, "rounds", rounds, os.environ["ORZ_FAST_K"], os.environ["ORZ_FAST_FAR"], flush=True)
    print("fuzz fast: %d cases, %d failures, %.0f s; worst size vs oracle per shape (inputs >= 100 KB): %s" %
          (cases, bad, time.time() - t0, {k: round(v, 2) for k, v in sorted(worst.items())}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
