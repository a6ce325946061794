"""dev: size and parse time of the fast mode over (tile, rounds) schedules, on the GPU -- the 100 MB text workload at -l1 / -l2 and
100 MB of zeros + 1 % noise at -l2, against the oracle's sizes (computed once per shape and level, in parallel processes)."""
import sys, os, time, json
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import corpus, orz_amd


def oracle_size(args):
    shape, level = args
    import _oracle
    d = corpus.enwik_like(100_000_000) if shape == "text" else corpus.zeros_noise(100_000_000)
    return shape, level, len(_oracle.encode(d, level))


if __name__ == "__main__":
    cases = [("text", 1), ("text", 2), ("zeros", 2)]
    with ProcessPoolExecutor(3) as ex:
        fut = ex.map(oracle_size, cases)
        data = {"text": corpus.enwik_like(100_000_000), "zeros": corpus.zeros_noise(100_000_000)}
        rows = []
        scheds = [(262144, 4), (262144, 3), (393216, 3), (524288, 3), (262144, 2), (524288, 2), (131072, 3)]
        if len(sys.argv) > 1:
            scheds = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
        for shape, level in cases:
            d = data[shape]
            for tile, rounds in scheds:
                enc = orz_amd.StreamEncoder(device=0, level=level, mode="fast", tile_bytes=tile, rounds=rounds)
                enc.encode(d[:20_000_000])
                out, st = enc.encode(d, stats=True)
                rows.append({"shape": shape, "level": level, "tile": tile, "rounds": rounds, "size": len(out), "ms": round(st["total_ms"], 1),
                             "parse_s": round(st["t_parse_s"], 3), "steps": st["sweeps"], "syncs": st["host_syncs"]})
                enc.close()
        ref = {(s, l): n for s, l, n in fut}
    for r in rows:
        r["delta_pct"] = round(100.0 * (r["size"] - ref[(r["shape"], r["level"])]) / ref[(r["shape"], r["level"])], 4)
        print(json.dumps(r), flush=True)
