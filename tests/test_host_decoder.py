"""The product's host decoder (orz_decode*, orz_lz_decoder_* in liborz_hip.so) and the `orz` CLI's decode
side, checked on CPU against streams produced by the oracle encoder."""
import ctypes
import os
import subprocess

import pytest

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "orz")


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("maker", ["text", "mixed", "zeros", "random", "p2"])
def test_decode_mem_matches_input(oracle, maker, level):
    import orz_amd

    n = 250_000
    data = {"text": lambda: _data.text(n), "mixed": lambda: _data.mixed(n), "zeros": lambda: _data.zeros_noise(n),
            "random": lambda: _data.random_bytes(n), "p2": lambda: _data.periodic(n, 2)}[maker]()
    stream = oracle.encode(data, level)
    out, used = orz_amd.decode_bytes(stream + b"\x07trailing bytes are ignored")
    assert out == data and used == len(stream)


@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
def test_decode_small(oracle, name):
    import orz_amd

    data = _data.SMALL_CASES[name]
    assert orz_amd.decode_bytes(oracle.encode(data, 2))[0] == data


def test_decode_across_block_slide(oracle):
    import orz_amd

    data = _data.mixed(16_777_216 + 123_456, seed=8)
    assert orz_amd.decode_bytes(oracle.encode(data, 0))[0] == data


def test_decode_rejects_garbage(oracle):
    import orz_amd

    good = oracle.encode(_data.text(40_000), 1)
    for bad in (good[: len(good) // 3], b"\x05abc", bytes([200, 200, 200, 200, 200, 200, 200, 200, 200, 200, 200])):
        with pytest.raises(Exception):
            orz_amd.decode_bytes(bad)


def test_object_level_decoder_call_pattern(oracle):
    """LZDecoder::decode chunk by chunk as orz::decode drives it (src/lib.rs:108-125)"""
    from orz_amd import _native

    lib = _native.load()
    data = _data.random_bytes(1_100_000) + _data.text(300_000)  # > 2^20 items -> two chunks
    stream = oracle.encode(data, 1)
    P, B = 16777215, (1 << 25) - 1
    win = (ctypes.c_uint8 * (2 * B + 960))()
    dec = lib.orz_lz_decoder_new()
    out = bytearray()
    at, spos, chunks = 0, P, 0
    while True:
        t, sh = 0, 0
        while True:
            b = stream[at]; at += 1
            t |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                break
        if t == 0:
            break
        end = ctypes.c_size_t()
        chunk = stream[at:at + t]
        rc = lib.orz_lz_decoder_decode(dec, chunk, t, ctypes.c_void_p(ctypes.addressof(win) + 480), spos, ctypes.byref(end))
        assert rc == 0
        out += bytes(win[480 + spos:480 + end.value])
        spos = end.value
        at += t
        chunks += 1
    lib.orz_lz_decoder_free(dec)
    assert chunks >= 2 and bytes(out) == data


def test_cli_decode_and_errors(oracle, tmp_path):
    assert os.path.exists(CLI), "bin/orz is built by __graft_entry__.build()"
    data = _data.mixed(120_000, seed=13)
    src, dst = tmp_path / "in.orz", tmp_path / "out.bin"
    src.write_bytes(oracle.encode(data, 2))
    subprocess.check_call([CLI, "decode", "-s", str(src), str(dst)])
    assert dst.read_bytes() == data
    # stdin/stdout defaults (src/main.rs:30-35)
    got = subprocess.run([CLI, "decode", "--silent"], input=src.read_bytes(), stdout=subprocess.PIPE, check=True).stdout
    assert got == data
    # invalid level -> error exit like the reference (src/main.rs:101)
    r = subprocess.run([CLI, "encode", "-l", "3", str(dst), str(tmp_path / "x")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"invalid level" in r.stderr
    r = subprocess.run([CLI, "decode", str(dst), str(tmp_path / "y")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"decoding failed" in r.stderr


def test_member_container_decode(oracle, tmp_path):
    """concatenated complete streams: decode_members / `orz decode --members` read all of them, the plain
    decoder stops after the first like the reference (src/lib.rs:108-110)"""
    import orz_amd

    parts = [_data.mixed(40_000, seed=100 + i) for i in range(3)] + [b""]
    container = b"".join(oracle.encode(p, 1) for p in parts)
    out, nm = orz_amd.decode_members(container)
    assert nm == 4 and out == b"".join(parts)
    assert orz_amd.decode_bytes(container)[0] == parts[0]
    src, dst = tmp_path / "m.orz", tmp_path / "m.out"
    src.write_bytes(container)
    subprocess.check_call([CLI, "decode", "-s", "--members", str(src), str(dst)])
    assert dst.read_bytes() == b"".join(parts)
    subprocess.check_call([CLI, "decode", "-s", str(src), str(dst)])
    assert dst.read_bytes() == parts[0]


def test_members_share_one_workspace(oracle):
    """orz_decode_members_mem reuses one window / model allocation for all members: a member that slid the
    window (> one block) must not leak history into the next one"""
    import orz_amd

    parts = [bytes(17_000_000), b"abc" * 1000, _data.mixed(50_000, seed=8), b"", _data.mixed(20_000, seed=9)]
    blob = b"".join(oracle.encode(p, i % 3) for i, p in enumerate(parts))
    out, m = orz_amd.decode_members(blob)
    assert m == len(parts) and out == b"".join(parts)
