"""Aggregate throughput of M independent members encoded concurrently on ONE GPU (one StreamEncoder +
HIP streams per member, driven by M host threads; ctypes releases the GIL inside the library)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import orz_amd  # noqa: E402


def run(members, nbytes, win, check=False):
    text = corpus.text_corpus(100_000_000)
    datas = [text[(i * 9_000_000) % (len(text) - nbytes):][:nbytes] for i in range(members)]
    encs = [orz_amd.StreamEncoder(0, 1) for _ in range(members)]
    for e in encs:
        e.set_tuning(62, win)
    outs = [None] * members

    def work(i):
        outs[i] = encs[i].encode(datas[i])

    for i in range(members):  # warm-up, sequential
        work(i)
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(members)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    ok = None
    if check:
        import _oracle
        ok = all(outs[i] == _oracle.encode(datas[i], 1) for i in range(members))
    for e in encs:
        e.close()
    return dict(members=members, bytes_each=nbytes, win=win, seconds=round(dt, 3), aggregate_MBps=round(members * nbytes / dt / 1e6, 2), exact=ok)


if __name__ == "__main__":
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 16_777_216
    for members, win in [(1, 2048), (2, 2048), (2, 1024), (4, 1024), (4, 512), (8, 512), (8, 256)]:
        print(json.dumps(run(members, nbytes, win, check=(members == 4 and win == 1024))), flush=True)
