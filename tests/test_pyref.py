"""Two independent restatements of the reference encoder must agree: the C oracle (oracle/) and the pure-Python
one written separately from the Rust source (tests/pyref/orz_py.py).  See that file's header for why."""
import os
import sys

import pytest

import _data

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyref"))
import orz_py  # noqa: E402

LEVELS = {0: (5, 3, 2), 1: (15, 9, 6), 2: (45, 27, 18)}  # src/main.rs:97-102


def test_known_answers_from_the_survey():
    """SURVEY.md A.8 (hand-derived from the source, independent of both restatements)"""
    assert orz_py.encode(b"", LEVELS[1]) == b"\x00"
    assert orz_py.encode(b"a", LEVELS[1]).hex() == "0c2aaaaaaaaaaa941eab40000000"


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
def test_small_cases_agree(oracle, name, level):
    data = _data.SMALL_CASES[name]
    assert orz_py.encode(data, LEVELS[level]) == oracle.encode(data, level)


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("maker", ["text", "mixed", "zeros", "random", "p1", "p2", "p3", "p5"])
def test_shapes_agree(oracle, maker, level):
    n = 12_000
    data = {"text": lambda: _data.text(n), "mixed": lambda: _data.mixed(n, seed=level + 1), "zeros": lambda: _data.zeros_noise(n),
            "random": lambda: _data.random_bytes(n), "p1": lambda: _data.periodic(n, 1), "p2": lambda: _data.periodic(n, 2),
            "p3": lambda: _data.periodic(n, 3), "p5": lambda: _data.periodic(n, 5)}[maker]()
    assert orz_py.encode(data, LEVELS[level]) == oracle.encode(data, level)


def test_longer_text_agrees(oracle):
    """enough items for ring thresholds, len_min updates and Huffman tables with many symbols"""
    data = _data.mixed(60_000, seed=42)
    assert orz_py.encode(data, LEVELS[1]) == oracle.encode(data, 1)


def test_one_context_fills_its_ring(oracle):
    """more than 4094 items in one context: ring recycling and stale hash heads (matcher.rs:176-179)"""
    data = (b"ab" * 3 + b"a%d;" % 7) * 1 + b"".join(b"a%03d" % (i * 7 % 1000) for i in range(6000))
    assert orz_py.encode(data, LEVELS[1]) == oracle.encode(data, 1)


@pytest.mark.parametrize("seed", range(24))
def test_random_structures_agree(oracle, seed):
    """random mixtures of text, repeats at random distances, runs, binary noise; random level"""
    import random

    rnd = random.Random(1000 + seed)
    base = _data.text(40_000, seed=seed + 2)
    out = bytearray()
    while len(out) < rnd.randrange(2_000, 9_000):
        kind = rnd.randrange(6)
        if kind == 0:
            at = rnd.randrange(len(base) - 400)
            out += base[at:at + rnd.randrange(1, 400)]
        elif kind == 1 and out:
            at = rnd.randrange(len(out))
            out += out[at:at + rnd.randrange(4, 300)]          # an earlier piece again
        elif kind == 2:
            out += bytes([rnd.randrange(256)]) * rnd.randrange(1, 600)
        elif kind == 3:
            out += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 200)))
        elif kind == 4:
            p = bytes(rnd.randrange(97, 123) for _ in range(rnd.randrange(1, 9)))
            out += p * rnd.randrange(2, 120)
        else:
            out += b" the " * rnd.randrange(1, 5) + base[:rnd.randrange(3, 40)]
    level = rnd.randrange(3)
    assert orz_py.encode(bytes(out), LEVELS[level]) == oracle.encode(bytes(out), level)


def test_two_blocks_agree(oracle):
    """window slide + LZEncoder::forward: a full 16 MiB block and a short second one, made of long runs and short text
    snippets so that pure Python gets through it; the second block matches into the history, and the last items of the
    first block are hashed with whatever lies past the block end (the hazard DESIGN.md section 2 describes)"""
    import random

    rnd = random.Random(77)
    base = _data.text(50_000, seed=5)
    out = bytearray()
    target = (1 << 24) + 300_000
    while len(out) < target:
        out += bytes([rnd.randrange(256)]) * rnd.randrange(400, 4000)
        at = rnd.randrange(len(base) - 64)
        out += base[at:at + rnd.randrange(8, 48)]
    # dense text on both sides of the block boundary: many items whose candidates sit in the slid history
    lo, hi = (1 << 24) - 120_000, (1 << 24) + 120_000
    k = 0
    while lo < hi:
        piece = base[(k * 7919) % 40_000:][:rnd.randrange(20, 900)]
        out[lo:lo + len(piece)] = piece
        lo += len(piece)
        k += 1
    data = bytes(out[:target])
    assert orz_py.encode(data, LEVELS[1]) == oracle.encode(data, 1)


def test_two_chunks_agree(oracle):
    """more than 2^20 items in one block: the second chunk starts without a census, with the running symbol ranks
    and fresh Huffman tables (lz.rs:131,238-270)"""
    data = _data.random_bytes(1_080_000)  # incompressible: one item per byte
    assert orz_py.encode(data, LEVELS[0]) == oracle.encode(data, 0)
