"""Every buffer of an encoder side by side (ORZ_ARENA_MB: one device allocation carved into the encoder's ~60 buffers, neighbours
instead of allocator slack behind each) -- the knob that exposed a round-1 out-of-bounds write of the exact parse at the end of
round 4 (62 words past a summary array on every full block: into slack with a hipMalloc per buffer, into the next array with
neighbours).  It lived in a dev script; VERDICT round 4 (weak 1c) asks for it in the GPU tier: one full-block case per parse mode.
An out-of-bounds write or a read of a neighbour changes bytes here that separate allocations hide."""
import pytest

import _data
import corpus

pytestmark = pytest.mark.gpu

N = (1 << 24) + 700_000  # a full block, the slide, a short block


def _encode(mode, data, level):
    import orz_amd

    enc = orz_amd.StreamEncoder(device=0, level=level, mode=mode)
    try:
        return enc.encode(data)
    finally:
        enc.close()


def test_exact_mode_full_block_with_buffers_side_by_side(oracle, monkeypatch):
    data = corpus.enwik_like(N)
    monkeypatch.setenv("ORZ_ARENA_MB", "6500")
    out = _encode("exact", data, 0)
    assert out == oracle.encode(data, 0)  # (the case that tripped the gate before the fix: 17 MB, -l0, a full first block)


def test_fast_mode_full_block_with_buffers_side_by_side(oracle, monkeypatch):
    data = corpus.enwik_like(N)
    plain = _encode("fast", data, 1)
    monkeypatch.setenv("ORZ_ARENA_MB", "6500")
    out = _encode("fast", data, 1)
    assert out == plain, "neighbouring buffers change the stream: %d vs %d bytes" % (len(out), len(plain))
    oracle.assert_decodes_to(out, data, "fast mode, buffers side by side")


def test_fast_mode_sparse_items_with_buffers_side_by_side(oracle, monkeypatch):
    data = _data.zeros_noise(N)  # few items, long matches: the other end of every per-item and per-position array
    plain = _encode("fast", data, 2)
    monkeypatch.setenv("ORZ_ARENA_MB", "6500")
    out = _encode("fast", data, 2)
    assert out == plain
    oracle.assert_decodes_to(out, data, "fast mode, zeros with noise, buffers side by side")
