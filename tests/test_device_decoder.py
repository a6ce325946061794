"""Device decoder (orz_amd/csrc/orz_decode_device.h): one member per wavefront, decoded by
one lane.  CPU tier: the kernel body and its host driver run on the emulation backend against containers made
by the oracle encoder.  GPU tier: the same through the C ABI (`orz_decode_members_device`), against the host
decoder and against members encoded on the GPU."""
import ctypes
import os
import subprocess

import pytest

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "orz")


@pytest.fixture(scope="module")
def emu_decode(emu):  # (the emu fixture builds build/libemu.so)
    lib = ctypes.CDLL(os.path.join(ROOT, "build", "libemu.so"))

    def decode(blob, slots=4):
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n, m = ctypes.c_size_t(), ctypes.c_size_t()
        err = ctypes.create_string_buffer(256)
        rc = lib.emu_decode_members(bytes(blob), ctypes.c_size_t(len(blob)), slots, ctypes.byref(dst), ctypes.byref(n),
                                    ctypes.byref(m), err, ctypes.c_size_t(256))
        if rc:
            raise ValueError(err.value.decode())
        out = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        return out, m.value

    return decode


def _parts():
    return [(_data.mixed(180_000, seed=1), 1), (_data.mixed(90_000, seed=2), 2), (b"", 1), (_data.zeros_noise(120_000), 2),
            (_data.random_bytes(40_000), 0), (_data.periodic(60_000, 3), 1), (b"x", 1), (_data.periodic(30_000, 1), 0),
            (_data.mixed(70_000, seed=9), 0)]


def test_emulated_kernel_decodes_a_mixed_container(emu_decode, oracle):
    parts = _parts()
    blob = b"".join(oracle.encode(p, lv) for p, lv in parts)
    out, m = emu_decode(blob, slots=4)  # 9 members through 4 state slots: three launches, state re-zeroed
    assert m == len(parts)
    assert out == b"".join(p for p, _ in parts)


def test_emulated_kernel_single_stream_and_empty(emu_decode, oracle):
    data = _data.mixed(300_000, seed=4)
    assert emu_decode(oracle.encode(data, 1)) == (data, 1)
    assert emu_decode(oracle.encode(b"", 1)) == (b"", 1)
    assert emu_decode(b"") == (b"", 0)


def test_emulated_kernel_multi_chunk_member(emu_decode, oracle):
    """a member with more than 2^20 items spans several chunks (fresh Huffman tables, running model state)"""
    data = _data.random_bytes(1_300_000)  # incompressible: one item per byte
    assert emu_decode(oracle.encode(data, 0)) == (data, 1)


def test_emulated_kernel_rejects_bad_containers(emu_decode, oracle):
    good = oracle.encode(_data.mixed(50_000, seed=3), 1)
    with pytest.raises(ValueError):
        emu_decode(good[:-1])          # EOF byte missing
    with pytest.raises(ValueError):
        emu_decode(good[: len(good) // 2])
    flipped = bytearray(good)
    for i in range(200, len(flipped), 997):
        flipped[i] ^= 0x5a
    try:  # corrupted payload: either reported, or decodes to something of the announced size -- never a crash
        emu_decode(bytes(flipped))
    except ValueError:
        pass


def test_emulated_kernel_decodes_members_of_several_blocks(emu_decode, oracle):
    """the window "slides" as a counter (src/lib.rs:119-124, src/matcher.rs:82-87): members beyond one block decode on the
    device decoder's kernel body too -- zeros (every match refers across the slide), and text of two blocks and a tail
    whose second block refers to sources in the first"""
    import corpus

    big = bytes(17_000_000)
    assert emu_decode(oracle.encode(big, 0)) == (big, 1)
    text = corpus.enwik_like(34_500_000)
    assert emu_decode(oracle.encode(text, 1) + oracle.encode(text[:70_000], 2)) == (text + text[:70_000], 2)


@pytest.mark.gpu
def test_gpu_decodes_what_the_oracle_encoded(oracle):
    import orz_amd

    parts = _parts()
    blob = b"".join(oracle.encode(p, lv) for p, lv in parts)
    out, m, st = orz_amd.decode_members_device(blob, stats=True)
    assert m == len(parts) and out == b"".join(p for p, _ in parts)
    assert st["members"] == len(parts) and st["launches"] == 1 and st["kernel_ms"] > 0
    assert (out, m) == orz_amd.decode_members(blob)


@pytest.mark.gpu
def test_gpu_round_trip_of_gpu_encoded_members():
    import orz_amd

    data = _data.mixed(6_000_000, seed=21)
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=2)
    blob, n = enc.encode(data, member_bytes=256 * 1024)
    enc.close()
    out, m = orz_amd.decode_members_device(blob)
    assert m == n == 23 and out == data


@pytest.mark.gpu
def test_gpu_many_members_in_batches(oracle, monkeypatch):
    """more members than state slots: several launches, blobs re-zeroed in between"""
    import orz_amd

    monkeypatch.setenv("ORZ_DECODE_SLOTS", "16")
    parts = [_data.mixed(3_000 + 97 * i, seed=i) for i in range(50)]
    blob = b"".join(oracle.encode(p, i % 3) for i, p in enumerate(parts))
    out, m, st = orz_amd.decode_members_device(blob, stats=True)
    assert m == 50 and st["launches"] == 4 and out == b"".join(parts)


@pytest.mark.gpu
def test_gpu_decoder_reports_bad_data(oracle):
    import orz_amd

    good = oracle.encode(_data.mixed(50_000, seed=3), 1)
    with pytest.raises(Exception):
        orz_amd.decode_members_device(good[:-1])


@pytest.mark.gpu
def test_gpu_decodes_the_default_64MiB_members():
    """what MemberEncoder writes by default (64 MiB members: four blocks, three slides) decodes on the GPU decoder"""
    import corpus
    import orz_amd

    data = corpus.enwik_like((1 << 26) + 3_000_000)  # (one lane decodes a member: ~2 MB/s each, so one full member and a short one)
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=2)
    try:
        blob, n = enc.encode(data)
    finally:
        enc.close()
    out, m = orz_amd.decode_members_device(blob)
    assert m == n == 2 and out == data


@pytest.mark.gpu
def test_cli_gpu_decode(oracle, tmp_path):
    parts = [_data.mixed(120_000, seed=5), _data.zeros_noise(80_000)]
    src = tmp_path / "c.orz"
    src.write_bytes(b"".join(oracle.encode(p, 1) for p in parts))
    dst = tmp_path / "c.out"
    r = subprocess.run([CLI, "decode", "-s", "--members", "--gpu", str(src), str(dst)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert dst.read_bytes() == b"".join(parts)


def test_size_claim_cap_leaves_the_most_compressible_members_alone(emu_decode, oracle):
    """index_members refuses members that announce more than 4096 bytes per byte of their own; runs of one byte are the
    densest thing an encoder writes (one 255-byte match per item of a few bits) and stay well below that"""
    for data in (bytes(3_000_000), b"ab" * 1_000_000):
        enc = oracle.encode(data, 2)
        assert len(data) < 4096 * len(enc)
        assert emu_decode(enc) == (data, 1)
