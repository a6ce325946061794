"""dev: does the per-block schedule cost size on multi-block inputs OTHER than the bench workload?  For each shape (40 MB: two full
blocks and a part) the stream's size with the schedule on / off against the oracle's encoder, -l1 and -l2, every stream through
the oracle's decoder."""
import json, os, sys
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import corpus, _data

N = 40_000_000


def make(shape):
    if shape == "mixed":
        return _data.mixed(N, seed=23)
    if shape == "synth_text":
        return _data.text(N, seed=31)
    if shape == "corpus_text":
        return corpus.text_corpus(N)
    if shape == "text_then_zeros":
        return corpus.enwik_like(20_000_000) + corpus.zeros_noise(20_000_000)
    if shape == "zeros_then_text":
        return corpus.zeros_noise(20_000_000) + corpus.enwik_like(20_000_000)
    raise SystemExit(shape)


def oracle_size(args):
    shape, level = args
    import _oracle
    return shape, level, len(_oracle.encode(make(shape), level))


if __name__ == "__main__":
    import orz_amd, _oracle
    shapes = sys.argv[1:] or ["mixed", "synth_text", "corpus_text", "text_then_zeros", "zeros_then_text"]
    cases = [(s, l) for s in shapes for l in (1, 2)]
    with ProcessPoolExecutor(10) as ex:
        fut = ex.map(oracle_size, cases)
        rows = []
        for shape in shapes:
            d = make(shape)
            for level in (1, 2):
                row = {"shape": shape, "level": level}
                for sched in ("on", "off"):
                    if sched == "off":
                        os.environ["ORZ_FAST_SCHED"] = "0"
                    else:
                        os.environ.pop("ORZ_FAST_SCHED", None)
                    enc = orz_amd.StreamEncoder(device=0, level=level)
                    out = enc.encode(d)
                    enc.close()
                    back, used = _oracle.decode(out)
                    row["size_" + sched] = len(out)
                    row["ok_" + sched] = bool(back == d and used == len(out))
                rows.append(row)
        ref = {(s, l): n for s, l, n in fut}
    for r in rows:
        o = ref[(r["shape"], r["level"])]
        r["oracle"] = o
        r["delta_on_pct"] = round(100.0 * (r["size_on"] - o) / o, 4)
        r["delta_off_pct"] = round(100.0 * (r["size_off"] - o) / o, 4)
        print(json.dumps(r), flush=True)
