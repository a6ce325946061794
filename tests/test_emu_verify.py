"""The validity gate (orz_amd/csrc/orz_verify.h) on the host emulation: every class of defect it names, injected after the parse
the way a parse defect would appear, fails the encode -- and a clean parse raises no finding (every other emulator test runs
with the gate on: it is the default).  The classes are the decoder's rules, LZDecoder::decode,
/root/reference/src/lz.rs:417-474: item sequence, contexts, ring membership and distance (src/matcher.rs:62-80), len_min and
the expected-length code (src/lz.rs:459-467), the words[] predictor (src/lz.rs:132-133,203,233)."""
import ctypes
import os

import pytest

import _data

CASES = {
    "hole": "hole/overlap in the item sequence",   # round 3's defect: an item rewritten, its span not re-parsed
    "context": "source in another ring",
    "ring": "source outside the ring",
    "lenmin": "length below len_min",
    "word": "WORD prediction",
    "bytes": "source bytes differ",
    # round 5: the len_min handed to ItemSyms damaged AFTER the parse computed it -- the gate keeps its own (it read that very value before)
    "lenmin2": "length code",
}


def _encode(emu, data):
    lib = emu.lib
    lib.emu_last_error.restype = ctypes.c_char_p
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    rc = lib.emu_encode_fast(data, ctypes.c_size_t(len(data)), 15, 9, 6, 0, 0, ctypes.byref(dst), ctypes.byref(n), None)
    if rc != 0:
        return None, lib.emu_last_error().decode()
    out = ctypes.string_at(dst, n.value)
    lib.emu_free(dst)
    return out, ""


@pytest.mark.parametrize("cls", sorted(CASES))
def test_gate_fails_the_encode_on_an_injected_defect(emu, oracle, cls, monkeypatch):
    data = _data.text(400_000, seed=5)
    monkeypatch.setenv("ORZ_VERIFY_INJECT", "%s:%d" % (cls, 3))
    out, err = _encode(emu, data)
    assert out is None, "the damaged parse went through (%d bytes)" % len(out)
    assert "validity gate" in err and CASES[cls] in err, err
    # (the clean parse of the same kind of input: every test of tests/test_emu_fast.py runs with the gate on)
