import sys, os, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import corpus, orz_amd
d = corpus.enwik_like(100_000_000)
for tile, rounds in [(131072, 4), (262144, 4), (524288, 4), (262144, 3), (131072, 3)]:
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="fast", tile_bytes=tile, rounds=rounds)
    enc.encode(d[:20_000_000])
    out, st = enc.encode(d, stats=True)
    print(json.dumps({"tile": tile, "rounds": rounds, "size": len(out), "ms": round(st["total_ms"], 1), "MBps": round(len(d) / 1e3 / st["total_ms"], 1),
                      "parse_s": round(st["t_parse_s"], 3), "post_s": round(st["t_post_s"], 3), "steps": st["sweeps"]}), flush=True)
    enc.close()
