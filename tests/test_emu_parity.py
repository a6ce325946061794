"""CPU tier of the parity tests: the device pipeline's kernel bodies (orz_amd/csrc), run on the host
emulation backend (tests/emu: thread kernels as loops, the wave-cooperative parse on a SIMT
emulator), must reproduce the oracle's stream byte for byte.  Sizes are small because the emulator
is slow; the GPU tier (test_gpu_parity.py) covers the full sizes through the C ABI."""
import os

import pytest

import _data

LEVELS = {0: (5, 3, 2), 1: (15, 9, 6), 2: (45, 27, 18)}


@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
def test_small_cases(emu, oracle, name):
    data = _data.SMALL_CASES[name]
    out, _ = emu(data)
    assert out == oracle.encode(data, 1)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_text_levels(emu, oracle, level):
    data = _data.text(30_000, seed=level + 1)
    out, st = emu(data, cfg=LEVELS[level])
    assert out == oracle.encode(data, level)
    assert st[1] >= 2  # it really iterated


@pytest.mark.parametrize("order", [0, 1, 2])
def test_sweep_order_does_not_matter(emu, oracle, order):
    """ascending = Gauss-Seidel, descending = pure Jacobi, shuffled: same fixed point"""
    data = _data.mixed(40_000, seed=11)
    out, _ = emu(data, order=order)
    assert out == oracle.encode(data, 1)


@pytest.mark.parametrize("order", [0, 2])
def test_late_waves_may_keep_their_previous_evaluation(emu, oracle, order, monkeypatch):
    """far-from-the-front waves that run late skip the sweep (test hook: about one evaluation in three, pseudo-random);
    the front never passes a skipped segment, so the fixed point is the same"""
    monkeypatch.setenv("ORZ_SKIP_RAND", "3")
    data = _data.mixed(30_000, seed=5)
    out, _ = emu(data, win=128, order=order)
    assert out == oracle.encode(data, 1)


@pytest.mark.parametrize("seg,win", [(62, 64), (62, 1024), (40, 256), (16, 512), (33, 128), (8, 600)])
def test_segment_and_window_sizes(emu, oracle, seg, win):
    data = _data.mixed(25_000 if seg > 16 else (8_000 if seg > 8 else 4_000), seed=seg)
    out, _ = emu(data, seg=seg, win=win)
    assert out == oracle.encode(data, 1)


@pytest.mark.parametrize("maker", ["zeros", "random", "p1", "p2", "p3", "p5"])
def test_degenerate_inputs(emu, oracle, maker):
    n = 20_000 if maker in ("zeros", "random") else 12_000  # (periodic data is the emulator's slowest case)
    data = {"zeros": lambda: _data.zeros_noise(n), "random": lambda: _data.random_bytes(n), "p1": lambda: _data.periodic(n, 1),
            "p2": lambda: _data.periodic(n, 2), "p3": lambda: _data.periodic(n, 3), "p5": lambda: _data.periodic(n, 5)}[maker]()
    out, _ = emu(data)
    assert out == oracle.encode(data, 1)


def test_ring_wraps_within_a_context(emu, oracle):
    # > 4094 items in one context: ring reuse, reduced offsets up to 4093 (src/matcher.rs:62-91)
    data = (b"ab " * 5000) + _data.text(12_000, seed=4) + (b"ab " * 1500)
    out, _ = emu(data)
    assert out == oracle.encode(data, 1)


def test_handoff_word_packs_and_unpacks(emu):
    """ExitPair (orz_parse.h): entry 27 bit, exit as a distance of < 512 positions + type, settled flag, sweep"""
    import ctypes
    import random

    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libemu.so"))
    lib.emu_exitpair.restype = ctypes.c_ulonglong
    out = (ctypes.c_uint * 4)()
    rnd = random.Random(5)
    kpre, kblock = 16_777_215, 33_554_431
    for _ in range(20_000):
        p = rnd.choice([kpre, kblock - 1, rnd.randrange(kpre, kblock)])
        d = rnd.choice([0, 1, 62, 240, 301, 511, rnd.randrange(0, 512)])
        if p + d > kblock:
            d = kblock - p
        entry, exit_ = (p << 2) | rnd.randrange(3), ((p + d) << 2) | rnd.randrange(3)
        sweep, settled = rnd.choice([0, 1, 4, (1 << 25) - 1, rnd.randrange(1 << 25)]), rnd.randrange(2)
        lib.emu_exitpair(sweep, settled, entry, exit_, out)
        assert list(out) == [entry, exit_, settled, sweep]


def test_huffman_kernel_matches_the_oracle(emu, oracle):
    """HuffWave (sort by counting, two-queue merge, depths by pointer jumping, ballot-counted canonical codes) on the host
    emulation == HuffmanTable::new_from_sym_weights + HuffmanEncoding (src/huffman.rs:27-141) for every table of
    tests/_huffcases.py: unused / single / tied symbols, weights that hit the 15-bit cap once and repeatedly"""
    import ctypes

    import numpy as np

    import _huffcases

    hw = _huffcases.weight_tables()
    hl, hc = _huffcases.oracle_tables(oracle, hw)
    assert hl.max() == 15 and any(hl[c].max() < 15 for c in range(hw.shape[0]))
    el, ec = np.zeros(hw.shape, np.uint8), np.zeros(hw.shape, np.uint16)
    emu.lib.emu_huff_build.restype = ctypes.c_int
    stride = emu.lib.emu_huff_build(hw.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(hw.shape[0]), el.ctypes.data_as(ctypes.c_void_p),
                                    ec.ctypes.data_as(ctypes.c_void_p))
    assert stride == _huffcases.STRIDE
    assert (el == hl).all()
    assert (ec == hc).all()
