"""The CPU oracle against everything that pins it: the hand-derived known answers of SURVEY.md A.8,
the reference's only unit test (src/coder.rs:224-265), the committed golden streams, and
encoder -> decoder round trips over the edge cases the domain has."""
import ctypes
import glob
import os

import numpy as np
import pytest

import _data

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answer_empty(oracle):
    for level in (0, 1, 2):
        assert oracle.encode(b"", level) == b"\x00"  # src/lib.rs:89: only the EOF chunk


def _varint_bits(v):  # src/coder.rs:27-38: per payload bit (more << 1) | bit, LSB first
    out = ""
    while True:
        more = v > 1
        out += ("1" if more else "0") + str(v & 1)
        v >>= 1
        if not more:
            return out


def test_known_answer_single_byte(oracle):
    # SURVEY.md A.8 derives this stream by hand from src/lz.rs:236-344 + src/coder.rs:27-67 (its
    # prose: 83 payload bits; the hex it prints has one "aa" too many).  The payload is rebuilt here
    # with an independent bit writer: varint(0) [no census symbols], varint(end_spos = 2^24),
    # varint(1 item), T0 empty, T1 = {max 1; symbol 96 (rank of 'a' with unlikely = 0) length 1},
    # T2 empty, the item's 1-bit code 0, zero pad to 32 bits; framed as LEB128(12) payload 0x00.
    bits = _varint_bits(0) + _varint_bits(1 << 24) + _varint_bits(1)
    bits += _varint_bits(0) + _varint_bits(0)                                    # T0: maxlen 0, end
    bits += _varint_bits(1) + _varint_bits(96 + 1) + _varint_bits(0) + _varint_bits(0)  # T1
    bits += _varint_bits(0) + _varint_bits(0)                                    # T2
    bits += "0"
    assert len(bits) == 83
    bits += "0" * (-len(bits) % 32)
    payload = int(bits, 2).to_bytes(len(bits) // 8, "big")
    want = bytes([len(payload)]) + payload + b"\x00"
    assert want.hex() == "0c2aaaaaaaaaaa941eab40000000"
    for level in (0, 1, 2):
        assert oracle.encode(b"a", level) == want


def test_reference_unit_test_string(oracle):
    # src/coder.rs:224-265: histogram -> Huffman(15) -> varint + table + symbols -> decode -> equal
    s = _data.SMALL_CASES["can"]
    buf = ctypes.create_string_buffer(4096)
    n = oracle.lib().orc_coder_selftest(s, ctypes.c_size_t(len(s)), buf, ctypes.c_size_t(4096))
    assert n > 0


def test_huffman_lengths_are_prefix_free_and_limited(oracle):
    rng = np.random.default_rng(0)
    L = oracle.lib()
    for trial in range(30):
        n = 389 if trial % 2 else 240
        w = (rng.zipf(1.3, n) * (rng.random(n) < 0.7)).astype(np.uint32)
        if trial == 0:
            w[:] = 0
        if trial == 1:
            w[:] = 0
            w[17] = 5
        if trial == 2:  # fibonacci-ish weights force the length limit (src/huffman.rs:98-108)
            a, b = 1, 1
            for i in range(40):
                w[i] = a
                a, b = b, a + b
        lens = (ctypes.c_uint8 * n)()
        mx = L.orc_huffman_lengths(w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_size_t(n), 15, lens)
        lens = np.frombuffer(lens, dtype=np.uint8)
        assert mx == lens.max() <= 15
        assert ((lens > 0) == (w > 0)).all()
        nz = lens[lens > 0]
        if len(nz) > 1:
            assert abs(sum(2.0 ** -int(x) for x in nz) - 1.0) < 1e-12  # Kraft equality
        elif len(nz) == 1:
            assert nz[0] == 1


def test_symrank_encode_decode_inverse(oracle):
    L = oracle.lib()
    enc, dec = oracle_symrank(L), oracle_symrank(L)
    rng = np.random.default_rng(4)
    for _ in range(5000):
        v, u = int(rng.integers(0, 389)), int(rng.integers(0, 256))
        r = L.orc_symrank_encode(ctypes.byref(enc), v, u)
        assert L.orc_symrank_decode(ctypes.byref(dec), r, u) == v


def oracle_symrank(L):
    class SR(ctypes.Structure):
        _fields_ = [("value", ctypes.c_uint16 * 389), ("index", ctypes.c_uint16 * 389), ("cnt", ctypes.c_uint32),
                    ("sum", ctypes.c_uint32)]

    s = SR()
    L.orc_symrank_new(ctypes.byref(s))
    ident = (ctypes.c_uint16 * 389)(*range(389))
    L.orc_symrank_init(ctypes.byref(s), ident)  # src/symrank.rs:31-36 (lz.rs:259-263 always inits before use)
    L.orc_symrank_encode.restype = ctypes.c_uint16
    L.orc_symrank_decode.restype = ctypes.c_uint16
    L.orc_symrank_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint16, ctypes.c_uint16]
    L.orc_symrank_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint16, ctypes.c_uint16]
    return s


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.orz"))))
def test_golden_streams(oracle, path):
    name, lvl, _ = os.path.basename(path).rsplit(".", 2)
    data = open(os.path.join(GOLD, name + ".in"), "rb").read()
    want = open(path, "rb").read()
    assert oracle.encode(data, int(lvl[1])) == want
    back, used = oracle.decode(want)
    assert back == data and used == len(want)


@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
@pytest.mark.parametrize("level", [0, 1, 2])
def test_roundtrip_small(oracle, name, level):
    data = _data.SMALL_CASES[name]
    back, _ = oracle.decode(oracle.encode(data, level))
    assert back == data


@pytest.mark.parametrize("maker", ["text", "zeros", "random", "mixed", "p1", "p2", "p3"])
def test_roundtrip_shapes(oracle, maker):
    n = 400_000
    data = {"text": lambda: _data.text(n), "zeros": lambda: _data.zeros_noise(n), "random": lambda: _data.random_bytes(n),
            "mixed": lambda: _data.mixed(n), "p1": lambda: _data.periodic(n, 1), "p2": lambda: _data.periodic(n, 2),
            "p3": lambda: _data.periodic(n, 3)}[maker]()
    for level in (0, 1, 2):
        enc = oracle.encode(data, level)
        back, used = oracle.decode(enc)
        assert back == data and used == len(enc)


def test_roundtrip_across_block_slide(oracle):
    # 16 MiB block boundary + a short final block (src/lib.rs:72-84, src/lz.rs:82-87)
    data = _data.mixed(16_777_216 + 300_000, seed=5)
    enc = oracle.encode(data, 0)
    back, _ = oracle.decode(enc)
    assert back == data


def test_more_than_one_chunk(oracle):
    # > 2^20 items in a block -> several chunks (src/lib.rs:32, src/lz.rs:128,131)
    data = _data.random_bytes(1_200_000)
    enc, items = oracle.encode(data, 1, trace_cap=len(data) + 8)
    assert len(items) > (1 << 20)
    back, _ = oracle.decode(enc)
    assert back == data


def test_decoder_stops_at_first_eof_chunk(oracle):
    # F7: concatenated streams -> only member 1 is decoded, `consumed` tells where it ended
    a, b = oracle.encode(b"hello hello hello", 1), oracle.encode(b"world", 1)
    back, used = oracle.decode(a + b)
    assert back == b"hello hello hello" and used == len(a)


def test_decoder_rejects_truncated(oracle):
    enc = oracle.encode(_data.text(5000), 1)
    with pytest.raises(ValueError):
        oracle.decode(enc[: len(enc) // 2])
