"""Deterministic stand-in corpora (enwik8/enwik9 are not available: SURVEY.md F3, section 8d).

T100  : first N bytes of the concatenation, in sorted path order, of text files (*.py *.rst *.md
        *.html *.txt, <= 2 MB each, */miopen/* excluded) under the image's dist-packages.
        Same image on the GPU box => same bytes; the SHA-256 is printed by bench.py.
Z     : zeros with 1 % uniform noise in 1..255, splitmix64 seed 0x6f727a (BASELINE config 4).
synth : pure-PRNG word-salad text, used when the dist-packages tree is missing.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = "/usr/local/lib/python3.10/dist-packages"
EXTS = (".py", ".rst", ".md", ".html", ".txt")
CACHE = os.environ.get("ORZ_CORPUS_CACHE", "/tmp/orz_corpus")


def _walk_sorted(root):
    paths = []
    for d, _dirs, files in os.walk(root):
        if "/miopen" in d:
            continue
        for f in files:
            if f.endswith(EXTS):
                paths.append(os.path.join(d, f))
    paths.sort()
    return paths


def synth_text(nbytes, seed=12345):
    """Word-salad with a Zipf-ish vocabulary: compressible, text-shaped, fully deterministic."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlucmfwypvbgkqjxz", dtype=np.uint8)
    vocab = []
    for _ in range(4096):
        n = int(rng.integers(2, 10))
        vocab.append(bytes(letters[np.minimum(rng.geometric(0.18, n) - 1, 25)]))
    ranks = np.minimum(rng.zipf(1.3, nbytes // 4 + 16) - 1, len(vocab) - 1)
    seps = [b" ", b" ", b" ", b" ", b", ", b". ", b"\n", b" the ", b" of "]
    sep_idx = rng.integers(0, len(seps), len(ranks))
    out = bytearray()
    for r, s in zip(ranks, sep_idx):
        out += vocab[r]
        out += seps[s]
        if len(out) >= nbytes:
            break
    return bytes(out[:nbytes])


def text_corpus(nbytes):
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "T%d.bin" % nbytes)
    if os.path.exists(path) and os.path.getsize(path) == nbytes:
        with open(path, "rb") as f:
            return f.read()
    buf = bytearray()
    if os.path.isdir(ROOT):
        for p in _walk_sorted(ROOT):
            try:
                if os.path.getsize(p) > 2 * 1024 * 1024:
                    continue
                with open(p, "rb") as f:
                    buf += f.read()
            except OSError:
                continue
            if len(buf) >= nbytes:
                break
    if len(buf) < nbytes:  # not enough text on this image: pad with synthetic text
        buf += synth_text(nbytes - len(buf))
    data = bytes(buf[:nbytes])
    tmp = "%s.%d.tmp" % (path, os.getpid())  # several ranks may build the cache at once: publish atomically
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)
    return data


def enwik_like(nbytes):
    """first `nbytes` of the enwik8-shaped workload (tools/enwik_like.py: committed word-level model + seeded
    counter-based sampling, so the bytes are the same on every box).  The canonical 100,000,000-byte text is cached
    under CACHE and its SHA-256 is checked against the value pinned in _gen.MANIFEST."""
    import enwik_like as _gen

    if nbytes <= 4_000_000:
        return _gen.generate(nbytes)
    full = 100_000_000
    if nbytes > full:  # larger inputs: whole copies + a prefix (the repeat distance is far beyond the 32 MiB window)
        base = enwik_like(full)
        return (base * (nbytes // full + 1))[:nbytes]
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "E%d.bin" % full)
    data = None
    if os.path.exists(path) and os.path.getsize(path) == full:
        with open(path, "rb") as f:
            data = f.read()
        if sha256(data) != _gen.MANIFEST[full]:
            data = None
    if data is None:
        data = _gen.generate(full)
        if sha256(data) != _gen.MANIFEST[full]:
            raise RuntimeError("enwik-like workload does not hash to the pinned value: %s" % sha256(data))
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)
    return data[:nbytes]


def _splitmix64(first, n, seed):
    """splitmix64 outputs number first+1 .. first+n of the stream seeded with `seed`"""
    x = (np.arange(first + 1, first + n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def zeros_noise(nbytes, seed=0x6F727A):
    """byte i = 0, except with probability 0.01 uniform in 1..255; generated in
    pieces so that a gigabyte does not need tens of gigabytes of temporaries"""
    out = np.zeros(nbytes, dtype=np.uint8)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for at in range(0, nbytes, step):
            n = min(step, nbytes - at)
            r = _splitmix64(at, n, seed)
            quarter = np.nonzero((r & np.uint64(3)) == 0)[0]            # r % 100 == 0  <=>  r % 4 == 0 and r % 25 == 0
            noisy = quarter[(r[quarter] % np.uint64(25)) == 0]
            out[at + noisy] = ((r[noisy] >> np.uint64(32)) % np.uint64(255) + np.uint64(1)).astype(np.uint8)
    return out.tobytes()


def sha256(data):
    return hashlib.sha256(data).hexdigest()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    d = text_corpus(n)
    print(len(d), sha256(d), os.path.join(CACHE, "T%d.bin" % n))
