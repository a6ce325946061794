"""CPU tier of the FAST parse mode: the kernel bodies of orz_amd/csrc/orz_fast.h run on the host emulation backend
(thread kernels as loops, the wave kernels on the SIMT emulator).  Parity for this mode: the stream decodes
bit-exactly with the oracle's decoder and its size stays close to the oracle encoder's; the GPU tier
(test_gpu_fast.py) holds the +-0.5 % band on the text workload at full size."""
import pytest

import _data

LEVELS = {0: (5, 3, 2), 1: (15, 9, 6), 2: (45, 27, 18)}


def _roundtrip(emu, oracle, data, level=1, band=None, **kw):
    out, st = emu.fast(data, cfg=LEVELS[level], **kw)
    back, used = oracle.decode(out)
    assert used == len(out) and back == data
    if band is not None:
        ref = oracle.encode(data, level)
        assert abs(len(out) - len(ref)) <= band * len(ref) + 64, (len(out), len(ref))
    return out, st


@pytest.mark.parametrize("name", sorted(_data.SMALL_CASES))
def test_small_cases(emu, oracle, name):
    _roundtrip(emu, oracle, _data.SMALL_CASES[name])


@pytest.mark.parametrize("level", [0, 1, 2])
def test_text_levels(emu, oracle, level):
    _roundtrip(emu, oracle, _data.text(60_000, seed=level + 1), level, band=0.02)


@pytest.mark.parametrize("maker", ["mixed", "zeros", "random", "p1", "p2", "p3", "p5"])
def test_data_shapes(emu, oracle, maker):
    n = 40_000
    data = {"mixed": lambda: _data.mixed(n, seed=11), "zeros": lambda: _data.zeros_noise(n), "random": lambda: _data.random_bytes(n),
            "p1": lambda: _data.periodic(n, 1), "p2": lambda: _data.periodic(n, 2), "p3": lambda: _data.periodic(n, 3),
            "p5": lambda: _data.periodic(n, 5)}[maker]()
    _roundtrip(emu, oracle, data, 1, band=0.03)


@pytest.mark.parametrize("tile,rounds", [(4096, 1), (4096, 4), (8192, 3), (65536, 8)])
def test_tile_and_round_settings(emu, oracle, tile, rounds):
    """any schedule yields a valid stream (the repairs see to that); more rounds / finer tiles only make it smaller"""
    _roundtrip(emu, oracle, _data.mixed(50_000, seed=3), 1, tile=tile, rounds=rounds)


@pytest.mark.parametrize("shape", ["empty", "one", "text", "mixed", "zeros", "random"])
def test_the_device_framed_stream_is_the_host_framed_stream(emu, oracle, shape):
    """round 6: FrameChunks / FrameAdvance / FrameEof frame { LEB128(t) chunk[t] }* + the EOF byte on the device
    (/root/reference/src/lib.rs:79-80,89, src/ioutil.rs:79-88) -- the same bytes the host framing of collect_one writes, into the
    encoder's own buffer and into a caller's; a caller's buffer that is too small fails the encode"""
    n = 150_000
    data = {"empty": lambda: b"", "one": lambda: b"a", "text": lambda: _data.text(n, seed=21), "mixed": lambda: _data.mixed(n, seed=22),
            "zeros": lambda: _data.zeros_noise(n), "random": lambda: _data.random_bytes(60_000)}[shape]()
    want, _ = emu.fast(data)
    own, err = emu.fast_device(data)
    assert err is None and own == want
    mine, err = emu.fast_device(data, cap=len(want))  # exactly the stream's size is enough
    assert err is None and mine == want
    back, used = oracle.decode(own)
    assert back == data and used == len(own)
    if len(want) > 1:
        none, err = emu.fast_device(data, cap=len(want) - 1)
        assert none is None and "too small" in err


def test_enwik_like_text_is_within_the_band(emu, oracle):
    import corpus

    data = corpus.enwik_like(1_500_000)
    out, st = _roundtrip(emu, oracle, data, 1, band=0.005)
    assert st[2] < st[3] // 100  # repairs are a small fraction of the items


def test_plan_encoder_accepts_what_it_wrote(oracle):
    """the oracle's plan-driven encoder on the oracle's own parse reproduces the oracle's stream (checker of the checker)"""
    import ctypes

    data = _data.mixed(80_000, seed=5)
    ref, tr = oracle.encode(data, 1, trace_cap=80_000)
    # the oracle's trace carries reduced offsets, not source positions: rebuild sources by replaying the rings
    heads = {}
    rings = {}
    plan = (oracle.PlanItem * len(tr))()
    P = oracle.P
    for i, it in enumerate(tr):
        pos = it.pos - P
        ctx = it.ctx & 0xff
        ring = rings.setdefault(ctx, [])
        plan[i].pos = pos
        if it.match_len:
            plan[i].type = 2
            plan[i].len = it.match_len
            plan[i].src = ring[len(ring) - 1 - it.reduced_offset]
        else:
            plan[i].type = 0 if it.symbol == 388 else 1
        ring.append(pos)
    assert oracle.encode_plan(data, (plan, len(tr))) == ref


def test_units_of_a_block(emu, oracle, monkeypatch):
    """the fast mode encodes a block in units that each close their last chunk early and slide the window by their own
    size (orz_stream.h, encode_block_units): the oracle's decoder -- the reference's loop, which slides only when the block
    is full -- must reproduce the input, and the size stays next to the single-unit encoding"""
    import corpus

    data = corpus.enwik_like(3_200_000)
    whole, _ = emu.fast(data, cfg=LEVELS[1])
    monkeypatch.setenv("ORZ_FAST_UNIT", str(1 << 20))
    out, st = emu.fast(data, cfg=LEVELS[1])
    assert st[0] == 3  # units encoded (the short rest joins the last one)
    back, used = oracle.decode(out)
    assert used == len(out) and back == data
    # (a unit's parse starts its tile pipeline anew: the first tiles of every unit see less than the whole block's do)
    assert abs(len(out) - len(whole)) <= 0.008 * len(whole)


def test_incremental_repair_passes_change_nothing(emu, oracle, monkeypatch):
    """after the first repair pass only the matches of runs that gained an item start are walked again (FastSource,
    orz_fast.h): the stream must be the very same as with every pass walking for every match -- on data and settings that
    need many repairs (one round, coarse tiles)"""
    import corpus

    cases = [(corpus.enwik_like(1_500_000), dict(tile=262144, rounds=1)), (_data.mixed(400_000, seed=7), dict(tile=131072, rounds=1))]
    for data, kw in cases:
        out, st = emu.fast(data, cfg=LEVELS[1], **kw)
        assert st[2] > 100  # repairs made
        monkeypatch.setenv("ORZ_FAST_FULLPASS", "1")
        full, st2 = emu.fast(data, cfg=LEVELS[1], **kw)
        monkeypatch.delenv("ORZ_FAST_FULLPASS")
        assert out == full and st[2] == st2[2]


def test_path_maps_too_large_for_lds_are_walked_in_global_memory(emu, oracle, monkeypatch):
    """PathTileDown stages the chunk maps of the active range in LDS when they fit (144 KB); eight rounds of 512 KiB
    tiles do not (1024 chunks x 240 entries): the same walks through global memory must give a valid stream of about
    the same size.  (ORZ_FAST_TDIV=1: the tile size asked for, although the input is short.)"""
    import corpus

    data = corpus.enwik_like(5_000_000)
    ref, _ = emu.fast(data, cfg=LEVELS[1])
    monkeypatch.setenv("ORZ_FAST_TDIV", "1")
    big, _ = emu.fast(data, cfg=LEVELS[1], tile=524288, rounds=8)
    back, used = oracle.decode(big)
    assert used == len(big) and back == data
    assert abs(len(big) - len(ref)) <= 0.006 * len(ref)


def test_deep_runs_take_older_candidates_from_the_compact_lists(emu, oracle):
    """deep runs take their older candidates from the compact lists of final item starts (FastListScan / FastRetire): the
    stream is valid and its size stays in the band on data whose runs are deep (zero runs with noise, text)"""
    import corpus

    _roundtrip(emu, oracle, _data.zeros_noise(300_000), 1, band=0.03)
    _roundtrip(emu, oracle, corpus.enwik_like(700_000), 1, band=0.005)


def test_a_reused_encoder_writes_what_a_fresh_one_writes(emu, oracle):
    """members are handed to whichever worker is free, so the bytes of a member must not depend on what its encoder
    encoded before.  The emulation fills the round state that no parse resets (ring horizons, dirty flags, remembered
    scan answers, evaluations) with garbage at every stream start: reading any of it before it is written shows here.
    (Round 3 found the ring horizons of the two positions a step evaluates beyond its newest tile left over from the
    stream before: valid streams, but a 64 MiB member came out 364 bytes different on a reused encoder.)"""
    import corpus

    text = corpus.enwik_like(6_000_000)
    for data in (text[3_000_000:], _data.mixed(700_000, seed=5), _data.zeros_noise(600_000)):
        fresh, _ = emu.fast(data, cfg=LEVELS[1])
        for first in (text[:2_500_000], _data.zeros_noise(500_000), b"abc"):
            again = emu.fast_reused(first, data, cfg=LEVELS[1])
            assert again == fresh
        assert oracle.decode(fresh)[0] == bytes(data)


def test_the_byte_before_the_window_travels_with_the_slide(emu, oracle):
    """The context of the history position at window offset 1 is hash1 of offset 0, which looks at the byte BEFORE the window
    (src/lz.rs:482-486).  The reference keeps a position in the bucket it was inserted into; the tables here are rebuilt
    from the window's bytes after every slide, so that byte has to travel with the slide -- with the front sentinel's zero
    in its place the position lands in the context without the letter-or-digit bit, a later position of that (rare) context
    takes it as a source, and no decoder finds it there (found by a soak of 64 MiB members: 5 of 640 invalid).  After a
    two-block stream the two bytes in front of the window must be the stream's: offset 0 is stream offset 1."""
    import ctypes

    key = b"QWERTYUIOPASDFGHJKLZ"
    head = b"a-" + key
    data = head + b"x" * ((1 << 24) - len(head)) + b"x" * 500 + b" -" + key + b"x" * 500
    front = (ctypes.c_uint8 * 2)()
    assert emu.lib.emu_window_front(data, ctypes.c_size_t(len(data)), front) == 0
    assert bytes(front) == b"\0a"  # (stream offset -1 is the sentinel, stream offset 0 the first byte)


def test_more_items_than_the_tail_buffers_start_with(emu, oracle):
    """incompressible input: an item per byte -- beyond the 6 M items the per-item buffers of a tail set hold to begin with
    (StreamEncoder::grow_tail_set); the stream must still decode"""
    data = _data.random_bytes(6_800_000)
    out, _ = emu.fast(data)
    assert oracle.decode(out)[0] == data


def test_buffers_side_by_side_change_nothing(emu, oracle, monkeypatch):
    """ORZ_EMU_ARENA_MB: every buffer of the encoder carved out of one block of pseudo-random bytes -- neighbours instead of
    slack behind each buffer, plausible garbage in what was not asked to be zeroed.  An out-of-bounds or uninitialised read
    changes the stream (round 4 found the exact parse's summary rebuild that way, at full block size); both modes must write
    what they write with a buffer of their own each."""
    data = _data.mixed(300_000, seed=17) + _data.text(300_000, seed=18)
    plain_fast, _ = emu.fast(data)
    monkeypatch.setenv("ORZ_EMU_ARENA_MB", "7000")
    side_fast, _ = emu.fast(data)
    side_exact, _ = emu(data[:120_000])
    assert side_fast == plain_fast
    assert side_exact == oracle.encode(data[:120_000], 1)  # (the full-block case of round 4: tools/dev, 57 minutes of CPU)


def test_failed_growth_of_the_tail_buffers_leaves_the_encoder_usable(emu, oracle, monkeypatch):
    """an allocation that fails half-way through grow_tail_set must leave the set as it was -- the encode fails,
    the same encoder then encodes the input like a fresh one (every new buffer is allocated before an old one is freed)."""
    import ctypes

    monkeypatch.setenv("ORZ_TAIL_ITEMS0", "65536")
    data = _data.random_bytes(200_000)  # one item per byte: more items than the buffers start with
    fresh, _ = emu.fast(data)
    for fail_in in (0, 5, 13):  # the first, a middle and the last allocation of the growth
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        rc = emu.lib.emu_encode_fast_after_failed_growth(data, ctypes.c_size_t(len(data)), 15, 9, 6, ctypes.c_long(fail_in), ctypes.byref(dst),
                                                         ctypes.byref(n))
        assert rc == 0, "fail_in=%d: rc %d" % (fail_in, rc)
        out = ctypes.string_at(dst, n.value)
        emu.lib.emu_free(dst)
        assert out == fresh
    back, used = oracle.decode(fresh)
    assert back == data and used == len(fresh)
