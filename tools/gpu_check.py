"""Quick on-GPU parity + timing probe (run through gpurun): HIP encoder vs oracle, byte for byte."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def frames(stream):
    """chunk payload lengths of an orz stream (src/lib.rs:108-110)"""
    out, at = [], 0
    while at < len(stream):
        t, sh = 0, 0
        while True:
            b = stream[at]; at += 1
            t |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                break
        if t == 0:
            break
        out.append(t); at += t
    return out


def main():
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1000, 1 << 20, 4 << 20]
    segs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64]
    wins = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8192]
    level = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    text = corpus.text_corpus(max(max(sizes), 1 << 20))
    enc = orz_amd.StreamEncoder(device=0, level=level, mode="exact")
    for n in sizes:
        data = text[:n]
        t0 = time.time()
        ref = _oracle.encode(data, level)
        t_or = time.time() - t0
        for seg in segs:
            for win in wins:
                enc.set_tuning(seg, win)
                t0 = time.time()
                out, st = enc.encode(data, stats=True)
                dt = time.time() - t0
                ok = out == ref
                rec = dict(n=n, seg=seg, win=win, level=level, equal=ok, out=len(out), ref=len(ref),
                           gpu_s=round(dt, 4), gpu_MBps=round(n / dt / 1e6, 2), oracle_MBps=round(n / t_or / 1e6, 2),
                           sweeps=st["sweeps"], seg_evals=st["seg_evals"], items=st["items"],
                           t_prep=round(st["t_prep_s"], 4), t_parse=round(st["t_parse_s"], 4),
                           t_post=round(st["t_post_s"], 4), parse_kernel_ms=round(st["parse_kernel_ms"], 3),
                           launches=st["parse_launches"])
                print(json.dumps(rec), flush=True)
                if not ok:
                    print("frames gpu:", frames(out)[:12], "ref:", frames(ref)[:12], flush=True)
                    i = next((i for i in range(min(len(out), len(ref))) if out[i] != ref[i]), None)
                    print("first differing byte:", i, flush=True)
                    dec = None
                    try:
                        dec = _oracle.decode(out)[0] == data
                    except Exception as e:  # noqa: BLE001
                        dec = repr(e)
                    print("MISMATCH: roundtrip via oracle decoder:", dec, flush=True)


if __name__ == "__main__":
    main()
