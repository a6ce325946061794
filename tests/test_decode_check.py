"""The engine of ORZ_VERIFY=decode (orz_amd/csrc/orz_decode_check.h) on the CPU: streams of the ORACLE's encoder, fed in pieces,
against their inputs -- the chunk-wise use of the library's host decoder (LZDecoder::decode, /root/reference/src/lz.rs:366-478;
the window slide of orz::decode, src/lib.rs:119-124) must accept what is right and name what is not."""
import ctypes

import pytest

import _data


@pytest.fixture(scope="module")
def check(emu):
    lib = emu.lib
    lib.emu_decode_check.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.emu_decode_check.restype = ctypes.c_int

    def run(stream, data, piece=0):
        err = ctypes.create_string_buffer(400)
        rc = lib.emu_decode_check(bytes(stream), len(stream), bytes(data), len(data), piece, err, 400)
        return rc, err.value.decode()

    return run


def test_streams_fed_in_pieces_across_a_window_slide(check, oracle):
    import corpus

    data = corpus.enwik_like(18_500_000)  # a full block and a part: the check slides its window like the decoder
    stream = oracle.encode(data, 1)
    for piece in (0, 777, 1 << 20):
        assert check(stream, data, piece) == (0, "")
    assert check(b"\x00", b"") == (0, "")
    small = _data.mixed(300_000, seed=4)
    assert check(oracle.encode(small, 0), small, 1) == (0, "")  # byte by byte


def test_what_is_wrong_is_named(check, oracle):
    data = _data.text(900_000, seed=2)
    stream = bytearray(oracle.encode(data, 1))
    bad = bytearray(stream)
    bad[len(bad) // 2] ^= 4
    rc, msg = check(bad, data, 4096)
    assert rc == 1 and "ORZ_VERIFY=decode" in msg and ("decodes to other bytes" in msg or "rejects a chunk" in msg), msg
    rc, msg = check(stream[:-1], data)  # the EOF byte is missing
    assert rc == 1 and "ends before" in msg, msg
    rc, msg = check(stream, data + b"x")  # more input than the stream encodes
    assert rc == 1 and "ends before" in msg, msg
    rc, msg = check(stream, data[:-1])  # less
    assert rc == 1, msg
    rc, msg = check(bytes(stream) + b"\x01", data)  # bytes behind the end
    assert rc == 1, msg
