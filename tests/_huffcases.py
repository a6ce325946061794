"""Weight tables for the Huffman kernel (HuffWave, orz_amd/csrc/orz_kernels.h) and what the ORACLE makes of them
(orc_huffman_lengths / orc_huffman_codes: src/huffman.rs:27-141).  Layout of one chunk as the encoder keeps it: 389 weights
of the symbol ranks after a match, 389 after a literal, 240 of the long match lengths (src/lz.rs:272-305)."""
import ctypes

import numpy as np

STRIDE = 389 * 2 + 240
TABLES = ((0, 389), (389, 389), (778, 240))


def weight_tables(seed=0):
    rng = np.random.default_rng(seed)
    tabs = []

    def fresh():
        return np.zeros(STRIDE, dtype=np.uint32)

    for trial in range(24):  # text-like: skewed, some symbols unused
        w = fresh()
        for off, n in TABLES:
            w[off:off + n] = (rng.zipf(1.2 + 0.1 * (trial % 5), n) * (rng.random(n) < 0.3 + 0.03 * trial)).clip(0, 1 << 20)
        tabs.append(w)
    w = fresh(); tabs.append(w.copy())                      # nothing used
    w = fresh(); w[17] = 5; w[389 + 388] = 1; w[778] = 9; tabs.append(w)   # one symbol per table
    w = fresh(); w[3] = 7; w[200] = 7; w[389] = 1; w[390] = 1 << 20; w[778 + 239] = 2; w[778 + 5] = 3; tabs.append(w)   # two symbols
    w = fresh(); w[:] = 1; tabs.append(w)                   # every symbol, all ties
    w = fresh(); w[:] = 1 << 12; w[::7] = (1 << 12) + 1; tabs.append(w)
    for k in range(4):  # fibonacci-ish weights force the length cap (src/huffman.rs:98-108), once or repeatedly
        w = fresh()
        for off, n in TABLES:
            a, b = 1, 1 + k
            idx = rng.permutation(n)
            for i in range(28 - 2 * k):
                w[off + idx[i]] = min(a, (1 << 20))
                a, b = b, a + b
            w[off + idx[40:80]] = rng.integers(0, 4, 40)
        tabs.append(w)
    w = fresh()                                              # powers of two and a crowd of ones
    for off, n in TABLES:
        w[off:off + n] = 1
        w[off:off + 20] = 1 << np.arange(20)
    tabs.append(w)
    for _ in range(6):  # dense, near-uniform: long ties between leaves and internal nodes
        w = fresh()
        w[:] = rng.integers(1, 4, STRIDE)
        tabs.append(w)
    for _ in range(6):
        w = fresh()
        w[:] = rng.integers(0, 1 << 20, STRIDE) * (rng.random(STRIDE) < 0.5)
        tabs.append(w)
    return np.ascontiguousarray(np.stack(tabs))


def oracle_tables(oracle, hw):
    """lengths [nchunks][STRIDE] u8 and codes u16 as the oracle builds them"""
    L = oracle.lib()
    hl = np.zeros(hw.shape, dtype=np.uint8)
    hc = np.zeros(hw.shape, dtype=np.uint16)
    for c in range(hw.shape[0]):
        for off, n in TABLES:
            w = np.ascontiguousarray(hw[c, off:off + n])
            lens = (ctypes.c_uint8 * n)()
            codes = (ctypes.c_uint16 * n)()
            L.orc_huffman_lengths(w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_size_t(n), 15, lens)
            L.orc_huffman_codes(lens, ctypes.c_size_t(n), codes)
            hl[c, off:off + n] = np.frombuffer(lens, dtype=np.uint8)
            hc[c, off:off + n] = np.frombuffer(codes, dtype=np.uint16)
    return hl, hc
