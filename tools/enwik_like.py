"""enwik8-shaped synthetic text (enwik8 itself is not in the tree: SURVEY.md F3).

A word-level Markov model (token = word + the separator that follows it; top-12 successors per token plus the
unigram distribution) trained ONCE on the prose of this image and committed as tools/enwik_model.npz, sampled
with a seeded counter-based generator.  The same bytes come out on every box: nothing but the committed model
and the seed enters.  The mixing parameters are calibrated so that the general-purpose compressors see what
they see on enwik8 (README.md:52-56 of the reference: gzip -6 36.5 %, bzip2 -9 29.0 %); the SHA-256 of the
standard sizes is pinned in MANIFEST below and checked by bench.py.

  python tools/enwik_like.py train   (re-creates the model from local prose; not needed to generate)
  python tools/enwik_like.py stats [nbytes]
"""
import hashlib
import os
import re
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL = os.path.join(HERE, "enwik_model.npz")
V = 24000        # tokens kept
S = 12           # successors per token

# calibrated mixing parameters (see `stats`)
P_UNIGRAM = 0.08   # probability of leaving the bigram table for the unigram distribution
GEO = 0.72         # successor rank ~ truncated geometric(GEO)
P_RARE = 0.21      # probability of an out-of-vocabulary word (random letters, Zipf length)
P_NUM = 0.012      # probability of a number token
P_LINK = 0.03      # probability of wrapping a word in [[ ]]

MANIFEST = {100_000_000: "cbbea047cfac1aa772cd46ad6db2d728fe49bbd0be9b154f89ef0e760b492ce5"}  # nbytes -> sha256 of generate(nbytes)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _uniform(step, lane, salt):
    """counter-based uniform [0,1) per (step, chain, purpose): no generator state, identical everywhere"""
    with np.errstate(over="ignore"):
        x = _splitmix(lane.astype(np.uint64) * np.uint64(0x100000001B3) + np.uint64(step) * np.uint64(0x9E3779B1) + np.uint64(salt) * np.uint64(0xD6E8FEB86659FD93))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def train(paths_root=("/usr/share/doc", "/usr/local/lib/python3.10/dist-packages", "/usr/share/common-licenses", "/usr/share/perl5", "/usr/share/man")):
    import gzip

    buf = bytearray()
    for root in paths_root:
        for d, _dirs, files in sorted(os.walk(root)):
            if "/miopen" in d:
                continue
            for f in sorted(files):
                p = os.path.join(d, f)
                try:
                    if f.endswith((".rst", ".md", ".txt", ".html", ".htm", ".pod", "README", "NEWS", "copyright")) and os.path.getsize(p) < 3_000_000:
                        buf += open(p, "rb").read()
                    elif f.endswith(".gz") and ("/man/" in p or "/doc/" in p) and os.path.getsize(p) < 500_000:
                        buf += gzip.open(p).read()
                except Exception:
                    pass
    text = bytes(b if (32 <= b < 127 or b == 10) else 32 for b in buf)
    toks = re.findall(rb"[A-Za-z0-9']+[^A-Za-z0-9']*", text)
    toks = [t if len(t) <= 24 else t[:24] for t in toks]
    from collections import Counter

    cnt = Counter(toks)
    vocab = [t for t, _ in cnt.most_common(V)]
    index = {t: i for i, t in enumerate(vocab)}
    ids = np.array([index.get(t, -1) for t in toks], dtype=np.int32)
    uni = np.array([cnt[t] for t in vocab], dtype=np.float64)
    succ = np.zeros((V, S), dtype=np.uint16)
    pair = Counter()
    prev = ids[:-1]
    nxt = ids[1:]
    ok = (prev >= 0) & (nxt >= 0)
    keys = prev[ok].astype(np.int64) * V + nxt[ok]
    uk, uc = np.unique(keys, return_counts=True)
    order = np.lexsort((-uc, uk // V))
    uk, uc = uk[order], uc[order]
    first = np.searchsorted(uk // V, np.arange(V))
    last = np.searchsorted(uk // V, np.arange(V), side="right")
    fallback = np.argsort(-uni)[:S].astype(np.uint16)
    for w in range(V):
        k = min(S, last[w] - first[w])
        succ[w, :k] = (uk[first[w]:first[w] + k] % V).astype(np.uint16)
        if k < S:
            succ[w, k:] = fallback[: S - k]
    blob = b"".join(vocab)
    lens = np.array([len(t) for t in vocab], dtype=np.uint8)
    np.savez_compressed(MODEL, blob=np.frombuffer(blob, dtype=np.uint8), lens=lens, uni=(uni / uni.sum()).astype(np.float32), succ=succ)
    print("model: %d tokens, %d bytes of vocabulary -> %s (%d bytes)" % (V, len(blob), MODEL, os.path.getsize(MODEL)))


_model = None


def _load():
    global _model
    if _model is None:
        m = np.load(MODEL)
        lens = m["lens"].astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        cum = np.cumsum(m["uni"].astype(np.float64))
        cum /= cum[-1]
        _model = (m["blob"], lens, starts, cum, m["succ"].astype(np.int64))
    return _model


CHAIN_BYTES = 65536  # every chain ("article") contributes exactly this many bytes, so generate(n) is a prefix of generate(m > n)


def _chains(first, count, seed):
    """bytes of chains first .. first+count-1 (each CHAIN_BYTES long), generated side by side"""
    blob, lens, starts, cum, succ = _load()
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    lane = np.arange(first, first + count, dtype=np.int64) + np.int64(seed % (1 << 31)) * 7919
    state = np.searchsorted(cum, _uniform(0, lane, 1)).clip(0, len(lens) - 1)
    g = GEO * (1 - GEO) ** np.arange(S)
    gcum = np.cumsum(g / g.sum())
    mean_tok = float((lens * np.diff(np.concatenate([[0.0], cum]))).sum()) + 0.6
    steps = int(CHAIN_BYTES / mean_tok * 1.3) + 64
    tok = np.empty((steps, count), dtype=np.int64)
    kind = np.zeros((steps, count), dtype=np.uint8)  # 0 vocab, 1 rare word, 2 number, +4 link
    aux = np.zeros((steps, count), dtype=np.uint64)
    for t in range(steps):
        u = _uniform(t + 1, lane, 2)
        rank = np.searchsorted(gcum, _uniform(t + 1, lane, 3)).clip(0, S - 1)
        nxt = succ[state, rank]
        uni_pick = np.searchsorted(cum, _uniform(t + 1, lane, 4)).clip(0, len(lens) - 1)
        nxt = np.where(u < P_UNIGRAM, uni_pick, nxt)
        r = _uniform(t + 1, lane, 5)
        k = np.where(r < P_RARE, 1, np.where(r < P_RARE + P_NUM, 2, 0)).astype(np.uint8)
        k |= (np.where(_uniform(t + 1, lane, 6) < P_LINK, 4, 0)).astype(np.uint8)
        tok[t] = nxt
        kind[t] = k
        aux[t] = (_uniform(t + 1, lane, 7) * float(1 << 52)).astype(np.uint64)
        state = nxt
    out = bytearray()
    for c in range(count):
        ids = tok[:, c]
        kd = kind[:, c]
        ax = aux[:, c]
        total = int(lens[ids].sum())
        idx = np.repeat(starts[ids] - np.concatenate([[0], np.cumsum(lens[ids])[:-1]]), lens[ids]) + np.arange(total)
        pb = blob[idx].tobytes()
        special = np.nonzero(kd)[0]
        offs = np.concatenate([[0], np.cumsum(lens[ids])])
        parts = []
        at = 0
        for s in special:  # splice the rare words / numbers / links in
            parts.append(pb[at:offs[s]])
            word = pb[offs[s]:offs[s + 1]]
            a = int(ax[s])
            core = word.rstrip(b" \n.,;:!?()-\"/")
            tail = word[len(core):] or b" "
            if kd[s] & 3 == 1:
                ln = 3 + (a % 7) + ((a >> 8) % 3)
                core = bytes(letters[[(a >> (5 * i)) % 26 if i < 10 else (a >> i) % 26 for i in range(ln)]])
                if (a >> 50) & 1:
                    core = core.capitalize()
            elif kd[s] & 3 == 2:
                core = str(a % (10 ** (1 + (a >> 40) % 4)) + (1800 if (a >> 45) & 1 else 0)).encode()
            if kd[s] & 4:
                core = b"[[" + core + b"]]"
            parts.append(core + tail)
            at = offs[s + 1]
        parts.append(pb[at:])
        chunk = b"".join(parts)
        assert len(chunk) >= CHAIN_BYTES, "chain too short"
        out += chunk[:CHAIN_BYTES]
    return out


def generate(nbytes, seed=0x656E77696B):
    nch = (nbytes + CHAIN_BYTES - 1) // CHAIN_BYTES
    out = bytearray()
    for first in range(0, nch, 512):
        out += _chains(first, min(512, nch - first), seed)
    return bytes(out[:nbytes])


def sha256(data):
    return hashlib.sha256(data).hexdigest()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        train()
    else:
        import bz2
        import time

        n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
        t = time.time()
        d = generate(n)
        print("generated %d bytes in %.1f s, sha256 %s" % (len(d), time.time() - t, sha256(d)))
        print("gzip -6 %.4f   bzip2 -9 %.4f   (enwik8: 0.365 / 0.290)" % (len(zlib.compress(d, 6)) / len(d), len(bz2.compress(d, 9)) / len(d)))
        print(d[:600].decode("latin1"))
