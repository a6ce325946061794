"""BASELINE.json's configurations at FULL size in the fast (default) mode, every one checked against the ORACLE:
the oracle's decoder (restatement of LZDecoder, /root/reference/src/lz.rs:366-478, driven like orz::decode,
src/lib.rs:94-129) must reproduce the input bit for bit, and the size must stay within +-0.5 % of the oracle's
encoder (LZEncoder::encode, src/lz.rs:131-346) at the same level ON THE SAME MEMBER SPLIT.

  configs[1]  100,000,000 bytes of enwik8-shaped text, -l1, one stream          (README.md:43-46 row "-l1")
  configs[2]  1 GB of text, -l2, independent members of 64 MiB                  (src/lib.rs:58-92 per member)
  configs[3]  the per-GPU shard of the 8 GB / 8 GPU job: 1 GB, -l1, members     (SURVEY.md 8d C3's single-GPU form)
  configs[4]  1 GB zeros + 1 % noise, -l2, members                              (ratio parity of the symrank / Huffman path)
  size table  the DEFAULT settings on text / mixed / zeros + noise / period 1, 3, 4, 7 at >= 4 MB, a stated band per shape

The oracle's work on 1 GB (15 members, encode + decode) runs on a thread pool: ctypes releases the GIL and the
members are independent streams.  Results are appended to gpurun_out/r06_configs_parity.jsonl when that directory
is writable (the builder copies the file into profiles/)."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import pytest

import _data

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SIZE_BAND = 0.005  # north_star: +-0.5 % of the reference at the same -l level
MEMBER = 1 << 26
GB = 1_000_000_000


def _record(row):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r06_configs_parity.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    print(json.dumps(row))


def _pool():
    return ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))


def _members_case(oracle, name, data, level, single_stream_delta):
    """encode `data` as 64 MiB members on one GPU (8 encoders), decode every member with the oracle, compare sizes with
    the oracle's encoder member by member"""
    import orz_amd
    from orz_amd import dist as od

    enc = orz_amd.MemberEncoder(device=0, level=level, jobs=8)
    try:
        t0 = time.time()
        container, nm = enc.encode(data, member_bytes=MEMBER)
        t_enc = time.time() - t0
    finally:
        enc.close()
    assert nm == (len(data) + MEMBER - 1) // MEMBER
    pieces = od.split_members(container)
    assert len(pieces) == nm
    spans = [(i * MEMBER, min(len(data), (i + 1) * MEMBER)) for i in range(nm)]
    def decode_one(k):
        try:
            return oracle.decode(pieces[k])
        except ValueError:  # keep the evidence: the member's index, its size and (when gpurun_out is writable) its bytes
            try:
                with open(os.path.join(ROOT, "gpurun_out", "bad_member_%d_of_%s.orz" % (k, name.split(":")[0].replace(" ", "_"))), "wb") as f:
                    f.write(pieces[k])
            except OSError:
                pass
            raise AssertionError("%s: the oracle's decoder rejects member %d (%d bytes; sizes %r)" % (name, k, len(pieces[k]), [len(p) for p in pieces]))

    with _pool() as ex:
        # every member through the ORACLE's decoder
        backs = list(ex.map(decode_one, range(len(pieces))))
        for (a, b), (back, used), piece in zip(spans, backs, pieces):
            assert used == len(piece)
            assert back == data[a:b]
        del backs
        # the oracle's encoder on the same member split
        refs = list(ex.map(lambda ab: len(oracle.encode(data[ab[0]:ab[1]], level)), spans))
    ref_total = sum(refs)
    delta = (len(container) - ref_total) / ref_total
    worst = max(abs(len(p) - r) / r for p, r in zip(pieces, refs))
    row = {"config": name, "bytes": len(data), "level": level, "members": nm, "member_bytes": MEMBER, "compressed": len(container),
           "oracle_same_split": ref_total, "delta_pct_same_split": round(100 * delta, 4), "worst_member_delta_pct": round(100 * worst, 4),
           "ratio": round(len(container) / len(data), 5), "encode_MBps_8_encoders_incl_host_copy": round(len(data) / t_enc / 1e6, 1),
           "oracle_decoder_round_trip": True}
    if single_stream_delta:
        t0 = time.time()
        one = len(oracle.encode(data, level))
        row["oracle_single_stream"] = one
        row["delta_pct_vs_single_stream"] = round(100 * (len(container) - one) / one, 4)
        row["oracle_single_stream_s"] = round(time.time() - t0, 1)
    _record(row)
    assert abs(delta) <= SIZE_BAND, row
    return row


def test_config1_full_size_fast_mode(oracle):
    """BASELINE configs[1] exactly as bench.py times it: the 100,000,000-byte workload, -l1, fast mode, one stream"""
    import corpus
    import orz_amd

    data = corpus.enwik_like(100_000_000)
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="fast")
    try:
        out = enc.encode(data)
    finally:
        enc.close()
    back, used = oracle.decode(out)
    assert used == len(out) and back == data
    ref = len(oracle.encode(data, 1))
    row = {"config": "C1: 100,000,000 bytes of enwik8-shaped text, -l1, fast mode, one stream", "compressed": len(out), "oracle": ref,
           "delta_pct": round(100 * (len(out) - ref) / ref, 4), "oracle_decoder_round_trip": True}
    _record(row)
    assert abs(len(out) - ref) <= SIZE_BAND * ref, row


def test_config2_text_1GB_l2_members(oracle):
    import corpus

    _members_case(oracle, "C2: 1 GB enwik8-shaped text, -l2, 64 MiB members, 8 encoders on one GPU", corpus.enwik_like(GB), 2, True)


def test_config3_shard_1GB_l1_members(oracle):
    import corpus

    _members_case(oracle, "C3 shard: 1 GB of the 8 GB text job (one GPU's share), -l1, 64 MiB members", corpus.enwik_like(GB), 1, False)


def test_config4_zeros_noise_1GB_l2_members(oracle):
    import corpus

    _members_case(oracle, "C4: 1 GB zeros + 1 % noise, -l2, 64 MiB members, 8 encoders on one GPU", corpus.zeros_noise(GB), 2, True)


# default settings, >= 4 MB per shape: how far ABOVE the oracle's encoder at -l1 each shape may come out (+ 64 bytes:
# degenerate shapes compress to ~3 KB).  Text must also stay within 0.5 % BELOW it (north_star: +-0.5 %); on the synthetic
# shapes the parse looks at more candidates than the reference's depth allows and comes out up to ~1 % smaller, which is
# recorded, not asserted against.
SHAPES = {
    "text": (lambda n: __import__("corpus").enwik_like(n), 0.005),
    "mixed": (lambda n: _data.mixed(n, seed=17), 0.005),
    "zeros_noise": (lambda n: _data.zeros_noise(n), 0.005),
    "period1": (lambda n: _data.periodic(n, 1), 0.005),
    "period3": (lambda n: _data.periodic(n, 3), 0.005),
    "period4": (lambda n: _data.periodic(n, 4), 0.005),
    "period7": (lambda n: _data.periodic(n, 7), 0.005),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_default_settings_size_table(oracle, shape):
    import orz_amd

    n = 6_000_000
    make, band = SHAPES[shape]
    data = make(n)
    enc = orz_amd.StreamEncoder(device=0, level=1)  # library defaults
    try:
        out = enc.encode(data)
    finally:
        enc.close()
    back, used = oracle.decode(out)
    assert used == len(out) and back == data
    ref = len(oracle.encode(data, 1))
    row = {"config": "default settings, -l1, %d bytes, shape %s" % (n, shape), "compressed": len(out), "oracle": ref,
           "delta_pct": round(100 * (len(out) - ref) / ref, 4), "delta_bytes": len(out) - ref, "band_pct": 100 * band}
    _record(row)
    assert len(out) - ref <= band * ref + 64, row
    # the low side of the band for every shape whose reference stream exceeds 64 KB; the period shapes
    # compress 6 MB to ~3.2 KB, where twenty bytes are 0.6 %: exempt, named as such in BASELINE.md
    if ref > 65536:
        assert ref - len(out) <= band * ref, row


# Rotations of the 100 MB workload whose 64 MiB members a soak (tools/dev/soak_members.py) found invalid before the window
# slide carried the byte in front of the history: rounds 7 and 9 failed deterministically (members 0; 1 and 5),
# the others in some runs only.  Eight members each, every one through the oracle's decoder, sizes against the oracle.
@pytest.mark.parametrize("rnd", [7, 9, 30, 40, 43])
def test_rotated_workload_members_from_the_soak(oracle, rnd):
    import corpus

    base = corpus.enwik_like(100_000_000)
    off = (rnd * 7_919_113) % (len(base) - 1)
    rot = base[off:] + base[:off]
    data = bytes((rot * ((8 * MEMBER) // len(rot) + 1))[: 8 * MEMBER])
    _members_case(oracle, "soak round %d: 8 members of 64 MiB from the workload rotated by %d, -l1" % (rnd, off), data, 1, False)
