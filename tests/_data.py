"""Seeded inputs shared by the CPU and GPU parity tests."""
import numpy as np

import corpus


def text(n, seed=1):
    return corpus.synth_text(n, seed=seed)


def zeros_noise(n):
    return corpus.zeros_noise(n)


def random_bytes(n, seed=3):
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()


def periodic(n, period):
    base = bytes((i * 37 + 11) % 251 for i in range(period))
    return (base * (n // period + 1))[:n]


def mixed(n, seed=9):
    """text with embedded runs, short periods and binary noise: exercises lazy matches, WORD and ring reuse"""
    rng = np.random.default_rng(seed)
    out = bytearray()
    t = corpus.synth_text(n, seed=seed)
    at = 0
    while len(out) < n:
        k = int(rng.integers(0, 6))
        ln = int(rng.integers(20, 2000))
        if k <= 2:
            out += t[at:at + ln]
            at = (at + ln) % max(1, len(t) - 4000)
        elif k == 3:
            out += bytes([int(rng.integers(0, 256))]) * ln
        elif k == 4:
            p = int(rng.integers(1, 7))
            out += periodic(ln, p)
        else:
            out += rng.integers(0, 256, ln // 4 + 1, dtype=np.uint8).tobytes()
    return bytes(out[:n])


SMALL_CASES = {
    "empty": b"",
    "one": b"a",
    "two": b"ab",
    "three": b"abc",
    "four": b"abcd",
    "run70": b"z" * 70,
    "abab": b"ab" * 200,
    "can": b"i can can a can into a can, can you can a can into a can?",  # the reference's unit-test string
    "sentence": b"the quick brown fox jumps over the lazy dog. " * 40,
}
