"""GPU tier: the HIP encoder, driven through the C ABI (liborz_hip.so via orz_amd), against the CPU
oracle -- bit-exact, on seeded inputs at sizes the oracle finishes in seconds, on the committed
golden streams, and at the BASELINE size through round trips with the oracle's decoder."""
import ctypes
import glob
import os

import pytest

import _data

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _exact_mode(monkeypatch):
    """this module is the byte-identical tier: every encoder built here runs the reference-identical parse"""
    monkeypatch.setenv("ORZ_MODE", "exact")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
@pytest.mark.parametrize("level", [0, 1, 2])
def test_small_cases(gpu_encoder_factory, oracle, name, level):
    data = _data.SMALL_CASES[name]
    assert gpu_encoder_factory(level).encode(data) == oracle.encode(data, level)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.orz"))))
def test_golden_streams(gpu_encoder_factory, path):
    name, lvl, _ = os.path.basename(path).rsplit(".", 2)
    data = open(os.path.join(GOLD, name + ".in"), "rb").read()
    assert gpu_encoder_factory(int(lvl[1])).encode(data) == open(path, "rb").read()


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("maker", ["text", "mixed", "zeros", "random", "p1", "p3"])
def test_shapes_bit_exact(gpu_encoder_factory, oracle, maker, level):
    n = 600_000
    data = {"text": lambda: _data.text(n), "mixed": lambda: _data.mixed(n), "zeros": lambda: _data.zeros_noise(n),
            "random": lambda: _data.random_bytes(n), "p1": lambda: _data.periodic(n, 1), "p3": lambda: _data.periodic(n, 3)}[maker]()
    assert gpu_encoder_factory(level).encode(data) == oracle.encode(data, level)


@pytest.mark.parametrize("seg,win", [(62, 512), (62, 8192), (32, 2048), (17, 1000)])
def test_tuning_does_not_change_the_stream(gpu_encoder_factory, oracle, seg, win):
    data = _data.mixed(1_000_000, seed=31)
    enc = gpu_encoder_factory(1)
    enc.set_tuning(seg, win)
    try:
        assert enc.encode(data) == oracle.encode(data, 1)
    finally:
        enc.set_tuning(62, 3072)


def test_more_than_one_chunk_per_block(gpu_encoder_factory, oracle):
    data = _data.random_bytes(1_300_000) + _data.text(400_000)
    out, st = gpu_encoder_factory(1).encode(data, stats=True)
    assert st["chunks"] >= 2
    assert out == oracle.encode(data, 1)


def test_block_slide_and_short_final_block(gpu_encoder_factory, oracle):
    # two full 16 MiB blocks + a short one: window slide, ring rebasing, tail-key hazard, stale tail bytes
    data = _data.mixed(2 * 16_777_216 + 234_567, seed=41)
    out, st = gpu_encoder_factory(1).encode(data, stats=True)
    assert st["blocks"] == 3
    assert out == oracle.encode(data, 1)


def test_item_trace_matches_oracle_parse(gpu_encoder_factory, oracle):
    """stage-level parity: every item's position, raw symbol, context, rank and length code"""
    data = _data.mixed(800_000, seed=51)
    enc = gpu_encoder_factory(1)
    enc.set_item_trace(True)
    try:
        out = enc.encode(data)
        tr = enc.item_trace()
    finally:
        enc.set_item_trace(False)
    ref, items = oracle.encode(data, 1, trace_cap=len(data) + 8)
    assert out == ref and len(tr) == len(items)
    for i in range(0, len(items), 97):
        o, g = items[i], tr[i]
        assert (g["pos"], g["symbol"], g["rank"], g["ctx"], g["enc_len"], g["unlikely"]) == (
            o.pos, o.symbol, o.rank, o.ctx, o.enc_len, o.unlikely)


def test_object_level_encoder_matches_reference_call_pattern(oracle):
    """LZEncoder::encode chunk by chunk + forward, exactly as orz::encode drives it (src/lib.rs:72-84)"""
    import orz_amd

    P, B, SENT = orz_amd.SBVEC_PREMATCH_LEN, orz_amd.LZ_BLOCK_SIZE, orz_amd.SBVEC_SENTINEL_LEN
    data = _data.random_bytes(1_150_000) + _data.mixed(16_777_216 - 1_150_000 + 500_000, seed=61)
    cfg = orz_amd.cfg_for_level(0)
    window = (ctypes.c_uint8 * (B + 2 * SENT))()
    enc = orz_amd.LZEncoder(device=0)
    stream = bytearray()
    off = 0
    while off < len(data):
        take = min(B - P, len(data) - off)
        ctypes.memmove(ctypes.addressof(window) + SENT + P, data[off:off + take], take)
        spos, sbuf_len = P, P + take
        nchunks = 0
        while spos < sbuf_len:
            spos, chunk = enc.encode(cfg, window, sbuf_len, spos)
            t = len(chunk)
            while t >= 128:
                stream.append(128 + t % 128)
                t //= 128
            stream.append(t)
            stream += chunk
            nchunks += 1
        if off == 0:
            assert nchunks >= 2
        off += take
        # sbvec.copy_within(len-P.., 0) ; lzenc.forward(len - P)
        ctypes.memmove(ctypes.addressof(window) + SENT, ctypes.addressof(window) + SENT + (B - P), P)
        enc.forward(B - P)
    stream.append(0)
    enc.close()
    assert bytes(stream) == oracle.encode(data, 0)


def test_stream_callbacks_api(oracle):
    import io

    import orz_amd

    data = _data.text(700_000, seed=71)
    src, dst = io.BytesIO(data), io.BytesIO()
    seen = []
    r, w = orz_amd.encode(src, dst, orz_amd.cfg_for_level(2), progress=lambda fin, a, b: seen.append((fin, a, b)))
    assert (r, w) == (len(data), len(dst.getvalue()))
    assert dst.getvalue() == oracle.encode(data, 2)
    assert seen and seen[-1][0] is True


def test_baseline_size_roundtrip_properties(gpu_encoder_factory, oracle):
    """100,000,000 bytes (BASELINE config 1 size): decode(encode(x)) == x through the oracle's decoder,
    encoding is deterministic, and the size equals the oracle encoder's (well inside the +-0.5 % band)."""
    import corpus

    data = corpus.text_corpus(100_000_000)
    enc = gpu_encoder_factory(1)
    out = enc.encode(data)
    back, used = oracle.decode(out)
    assert used == len(out) and back == data
    ref = oracle.encode(data, 1)
    assert abs(len(out) - len(ref)) <= 0.005 * len(ref)
    assert out == ref


def test_cli_encode_roundtrip(oracle, tmp_path):
    """`orz encode -l1 in out` (GPU) -> reference-format stream: equals the oracle's, and `orz decode` restores it"""
    import subprocess

    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "orz")
    data = _data.mixed(900_000, seed=81)
    src, enc, dec = tmp_path / "in.bin", tmp_path / "in.orz", tmp_path / "out.bin"
    src.write_bytes(data)
    subprocess.check_call([cli, "encode", "-s", "-l1", str(src), str(enc)])
    assert enc.read_bytes() == oracle.encode(data, 1)
    subprocess.check_call([cli, "decode", "-s", str(enc), str(dec)])
    assert dec.read_bytes() == data


def test_gpu_stream_decodes_with_product_decoder(gpu_encoder_factory):
    import orz_amd

    data = _data.text(2_000_000, seed=91)
    out = gpu_encoder_factory(2).encode(data)
    assert orz_amd.decode_bytes(out)[0] == data


def test_members_concurrent_on_one_gpu(oracle, tmp_path):
    """independent members encoded concurrently (3 encoders, one GPU): every member is exactly the stream the
    oracle produces for that slice, and the container decodes back to the input"""
    import subprocess

    import orz_amd
    from orz_amd import dist as od

    data = _data.mixed(22_000_000, seed=111)
    member = 6_000_000
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=3)
    try:
        container, nm = enc.encode(data, member_bytes=member)
    finally:
        enc.close()
    pieces = od.split_members(container)
    assert nm == len(pieces) == 4
    for i, p in enumerate(pieces):
        assert p == oracle.encode(data[i * member:(i + 1) * member], 1)
    assert orz_amd.decode_members(container) == (data, 4)
    # the same through the command line
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "orz")
    src, encf, decf = tmp_path / "in.bin", tmp_path / "in.orz", tmp_path / "out.bin"
    src.write_bytes(data[:9_000_000])
    subprocess.check_call([cli, "encode", "-s", "-l1", "--member-size", "4000000", "--jobs", "2", str(src), str(encf)])
    subprocess.check_call([cli, "decode", "-s", "--members", str(encf), str(decf)])
    assert decf.read_bytes() == data[:9_000_000]


@pytest.mark.parametrize("kind", ["text", "zeros_noise"])
def test_baseline_configs_2_and_4_members_at_l2(oracle, kind):
    """BASELINE.json configs[2] / configs[4] in small: -l2, full 16 MiB members (the sweep window of -l2 is
    sized from the kernel's LDS footprint), text and the degenerate zeros + 1 % noise regime; every member is the
    oracle's stream for that slice.  tools/gpu_configs.py runs the same at 1 GB (round trip only)."""
    import corpus
    import orz_amd
    from orz_amd import dist as od

    member = 1 << 24
    n = 2 * member + 3_000_000
    data = corpus.text_corpus(100_000_000)[5_000_000:5_000_000 + n] if kind == "text" else corpus.zeros_noise(n)
    enc = orz_amd.MemberEncoder(device=0, level=2, jobs=2)
    try:
        container, nm = enc.encode(data, member_bytes=member)
    finally:
        enc.close()
    pieces = od.split_members(container)
    assert nm == len(pieces) == 3
    for i, p in enumerate(pieces):
        assert p == oracle.encode(data[i * member:(i + 1) * member], 2), "member %d" % i
    assert orz_amd.decode_members(container) == (data, 3)


def test_randomised_settings_do_not_change_the_stream():
    """short run of tools/gpu_fuzz.py: random shapes / sizes / levels / speculation settings vs the oracle"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "12", "7"], capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout


def test_huffman_kernel_matches_the_oracle(oracle):
    """orz_huffman_tables (the encoder's HuffWave kernel alone, through the C ABI) == the oracle's
    HuffmanTable::new_from_sym_weights + HuffmanEncoding (src/huffman.rs:27-141) on every table of tests/_huffcases.py;
    weights the keys cannot hold are refused.  The launch time goes to gpurun_out/r03_huffman_kernel.json."""
    import json

    import numpy as np

    import _huffcases
    import orz_amd

    hw = _huffcases.weight_tables()
    hl, hc = _huffcases.oracle_tables(oracle, hw)
    lens, codes, us = orz_amd.huffman_tables(hw)
    assert (lens == hl).all()
    assert (codes == hc).all()
    # the encoder's own shape: the 5 chunks (15 tables) of a 16 MiB block of text
    five = np.ascontiguousarray(hw[:5])
    l5, c5, us5 = orz_amd.huffman_tables(five)
    assert (l5 == hl[:5]).all() and (c5 == hc[:5]).all()
    row = {"kernel": "HuffWave", "tables": int(hw.shape[0]) * 3, "launch_us": round(us, 1), "tables_of_one_block": 15, "launch_us_one_block": round(us5, 1)}
    print(json.dumps(row))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r03_huffman_kernel.json"), "w") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    bad = hw.copy()
    bad[3, 7] = 1 << 23
    with pytest.raises(Exception):
        orz_amd.huffman_tables(bad)
