"""The product's validity gates on the GPU (task 2).
 * the per-block gate (orz_amd/csrc/orz_verify.h, on by default): each class of defect, injected after the parse, fails the encode
   (same injections as tests/test_emu_verify.py on the emulation);
 * ORZ_VERIFY=decode / `orz encode --verify`: the finished stream through the library's own decoder before it leaves the library
   -- a bit flipped BEHIND the per-block gate (in the packed bytes) fails the encode; clean streams pass, members too.
Rules checked: LZDecoder::decode, /root/reference/src/lz.rs:417-474."""
import os
import subprocess
import sys

import pytest

import _data

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "hole": "hole/overlap in the item sequence",
    "context": "source in another ring",
    "ring": "source outside the ring",
    "lenmin": "length below len_min",
    "word": "WORD prediction",
    "bytes": "source bytes differ",
    # round 5: the len_min handed to ItemSyms damaged AFTER the parse computed it -- the gate keeps its own (it read that very value before)
    "lenmin2": "length code",
}


@pytest.mark.parametrize("cls", sorted(CASES))
def test_gate_fails_the_encode_on_an_injected_defect(oracle, cls, monkeypatch):
    import orz_amd

    data = _data.text(6_000_000, seed=5)
    monkeypatch.setenv("ORZ_VERIFY_INJECT", "%s:%d" % (cls, 11))
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="fast")
    try:
        with pytest.raises(orz_amd.OrzError) as ei:
            enc.encode(data)
    finally:
        enc.close()
    assert "validity gate" in str(ei.value) and CASES[cls] in str(ei.value), str(ei.value)
    monkeypatch.delenv("ORZ_VERIFY_INJECT")
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="fast")
    try:
        out = enc.encode(data)
    finally:
        enc.close()
    assert oracle.decode(out)[0] == data


def test_gate_also_guards_the_exact_mode(oracle, monkeypatch):
    import orz_amd

    data = _data.text(2_000_000, seed=6)
    monkeypatch.setenv("ORZ_VERIFY_INJECT", "hole:3")
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="exact")
    try:
        with pytest.raises(orz_amd.OrzError) as ei:
            enc.encode(data)
    finally:
        enc.close()
    assert "validity gate" in str(ei.value)


def test_decode_verification_catches_what_the_gate_cannot_see(oracle, monkeypatch):
    import orz_amd

    data = _data.text(3_000_000, seed=8)
    monkeypatch.setenv("ORZ_VERIFY", "decode")
    enc = orz_amd.StreamEncoder(device=0, level=1)
    try:
        out = enc.encode(data)  # clean: passes, and the oracle agrees
    finally:
        enc.close()
    assert oracle.decode(out)[0] == data
    monkeypatch.setenv("ORZ_OUTPUT_INJECT", "70000")  # one bit of the first chunk, flipped behind the per-block gate
    enc = orz_amd.StreamEncoder(device=0, level=1)
    try:
        with pytest.raises(orz_amd.OrzError) as ei:
            enc.encode(data)
    finally:
        enc.close()
    assert "ORZ_VERIFY=decode" in str(ei.value)
    monkeypatch.delenv("ORZ_VERIFY")  # without the decode check the damaged stream leaves the library: the check is what catches it
    enc = orz_amd.StreamEncoder(device=0, level=1)
    try:
        bad = enc.encode(data)
    finally:
        enc.close()
    with pytest.raises(Exception):
        back, _ = oracle.decode(bad)
        assert back == data


def test_decode_verification_of_members_and_of_the_command_line(oracle, tmp_path, monkeypatch):
    import orz_amd

    data = _data.text(40 << 20, seed=9)
    monkeypatch.setenv("ORZ_VERIFY", "decode")
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=4)
    try:
        blob, nm = enc.encode(data, member_bytes=20 << 20)  # two blocks per member: the check slides its window too
    finally:
        enc.close()
    assert nm == 2 and orz_amd.decode_members(blob)[0] == data
    monkeypatch.delenv("ORZ_VERIFY")
    src, dst = tmp_path / "in.bin", tmp_path / "out.orz"
    src.write_bytes(data[: 18 << 20])
    exe = os.path.join(ROOT, "bin", "orz")
    r = subprocess.run([exe, "encode", "-s", "-l1", "--verify", str(src), str(dst)], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert oracle.decode(dst.read_bytes())[0] == data[: 18 << 20]
    env = dict(os.environ, ORZ_OUTPUT_INJECT="70000")
    r = subprocess.run([exe, "encode", "-s", "-l1", "--verify", str(src), str(dst)], stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode != 0 and "ORZ_VERIFY=decode" in r.stderr, r.stderr
