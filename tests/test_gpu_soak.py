"""Soak tier of the GPU tests (task 1): the members path under the conditions that exposed the round-3 defect --
eight encoders at once on one GPU, fresh and reused, inputs cut from rotations of the workload -- with EVERY distinct stream
through the ORACLE's decoder (restatement of LZDecoder::decode, /root/reference/src/lz.rs:366-478, driven like orz::decode,
src/lib.rs:94-129), and the GPU's bytes compared with the host emulation's (tests/golden/emu_soak.json, made by
tests/golden/make_emu_soak.py on the CPU: the emulation runs a launch's threads one after another, so equality says the
concurrent execution on the GPU changed nothing).

The defect these tests pin: FastWordCheck rewrote a WORD item in the launch that judged it; the thread of
the next position, in another wavefront, could see the new item-start bit next to the round's stale decision and cut the match
behind it to one byte -- a hole in the path.  It needed a WORD item in a wavefront's last lane and a busy GPU: ~30 % of the
eight-encoder runs of soak round 40 wrote one or two undecodable members of 64 MiB, a lone encoder never did."""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
MEMBER = 1 << 26


def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "emu_soak.json")) as f:
        return json.load(f)


def _rotation(base, rnd, nbytes):
    off = (rnd * 7_919_113) % (len(base) - 1)
    rot = base[off:] + base[:off]
    return bytes((rot * (nbytes // len(rot) + 1))[:nbytes])


def _check_streams(oracle, data, member_bytes, pieces, what):
    """every stream of `pieces` (member k = data[k * member_bytes : ...]) through the oracle's decoder"""
    def one(k):
        exp = data[k * member_bytes:(k + 1) * member_bytes]
        try:
            back, used = oracle.decode(pieces[k])
            ok = used == len(pieces[k]) and back == exp
        except ValueError:
            ok = False
        if not ok:  # name the first item no decoder follows (oracle/orz_diag.c) and keep the stream
            d = oracle.diag(pieces[k], exp)
            try:
                with open(os.path.join(ROOT, "gpurun_out", "soak_bad_%s_m%d.orz" % (what.replace(" ", "_"), k)), "wb") as f:
                    f.write(pieces[k])
            except OSError:
                pass
            return "%s: member %d (%d bytes) does not decode to its input; first wrong item: %r" % (
                what, k, len(pieces[k]), {x: d[x] for x in d if not x.startswith("near_") and x != "tab_near"})
        return None

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
        bad = [r for r in ex.map(one, range(len(pieces))) if r]
    assert not bad, "\n".join(bad)


def test_word_repair_race_reproducer(oracle):
    """The first 17 MiB of member 5 of soak round 40, eight copies through eight encoders, six times: before the fix 22 of 96
    such streams differed from the others (28 bytes shorter: a literal at stream offset 1,846,529 in place of a match of 44
    bytes, then nothing for 43 positions).  All 48 must equal the host emulation's stream, which the oracle decodes."""
    import corpus
    import orz_amd
    from orz_amd import dist as od

    n = 17 * (1 << 20)
    base = corpus.enwik_like(100_000_000)
    piece = _rotation(base, 40, 8 * MEMBER)[5 * MEMBER:5 * MEMBER + n]
    want = _golden()["round 40 member 5, first 17 MiB"]
    assert hashlib.sha256(piece).hexdigest() == want["input_sha256"]
    seen = {}
    for rep in range(6):
        enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
        try:
            blob, nm = enc.encode(piece * 8, member_bytes=n)
        finally:
            enc.close()
        assert nm == 8
        for p in od.split_members(blob):
            h = hashlib.sha256(p).hexdigest()
            seen.setdefault(h, p)
    assert list(seen) == [want["sha256"]], "streams written: %r, the emulation's: %s" % ({h: len(p) for h, p in seen.items()}, want["sha256"])
    back, used = oracle.decode(seen[want["sha256"]])
    assert back == piece and used == want["bytes"]


def test_gpu_bytes_equal_the_emulation_at_64MiB_fresh_and_reused(oracle):
    """parity bar (4) at full member size: the eight 64 MiB members of soak round 40 from eight concurrent encoders -- fresh
    ones, the same ones again, and the same ones after they encoded another rotation -- are byte for byte what the host
    emulation of the same kernels writes (members 4 and 5 are the two the round-3 driver run found undecodable)"""
    import corpus
    import orz_amd
    from orz_amd import dist as od

    gold = _golden()
    base = corpus.enwik_like(100_000_000)
    data = _rotation(base, 40, 8 * MEMBER)
    other = _rotation(base, 41, 8 * MEMBER)
    want = [gold["round 40 member %d" % k]["sha256"] for k in range(8)]
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
    try:
        runs = []
        for what, d in (("fresh", data), ("reused", data), ("other", other), ("reused after another input", data)):
            blob, nm = enc.encode(d, member_bytes=MEMBER)
            if d is data:
                runs.append((what, od.split_members(blob)))
    finally:
        enc.close()
    for what, pieces in runs:
        got = [hashlib.sha256(p).hexdigest() for p in pieces]
        assert got == want, "%s encoders: members %r differ from the emulation" % (what, [k for k in range(8) if got[k] != want[k]])
    _check_streams(oracle, data, MEMBER, runs[0][1], "round 40")


def test_soak_rotations_every_member_through_the_oracle(oracle):
    """64 rotations of the workload x 512 MiB, cut into members of 64 MiB (4 blocks), 20 MiB (a full block and a short one:
    the window slides) and 10 MiB in turn -- 2,208 members, 34 GB -- through ONE set of eight encoders that is never rebuilt;
    every eighth rotation a fresh set encodes the same input and must write the same bytes.  Every distinct stream goes
    through the oracle's decoder while the GPU encodes the next rotation.  0 invalid, 0 differences."""
    import corpus
    import orz_amd
    from orz_amd import dist as od

    rotations = int(os.environ.get("ORZ_SOAK_ROTATIONS", "64"))
    base = corpus.enwik_like(100_000_000)
    sizes = [MEMBER, 20 << 20, 10 << 20, 10 << 20]
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
    pool = ThreadPoolExecutor(max_workers=1)  # one rotation is checked (on its own thread pool) while the next is encoded
    pending = None
    members = nbytes = 0
    t0 = time.time()
    t_enc = 0.0
    try:
        for rnd in range(rotations):
            mb = sizes[rnd % 4]
            data = _rotation(base, rnd, 8 * MEMBER)
            t1 = time.time()
            blob, nm = enc.encode(data, member_bytes=mb)
            t_enc += time.time() - t1
            pieces = od.split_members(blob)
            assert len(pieces) == nm == (len(data) + mb - 1) // mb
            if rnd % 8 == 0:
                fresh = orz_amd.MemberEncoder(device=0, level=1, jobs=8)
                try:
                    blob2, _ = fresh.encode(data, member_bytes=mb)
                finally:
                    fresh.close()
                assert blob2 == blob, "rotation %d: fresh encoders write other bytes than the reused ones (members %r)" % (
                    rnd, [k for k, (a, b) in enumerate(zip(od.split_members(blob2), pieces)) if a != b])
            if pending is not None:
                pending.result()
            pending = pool.submit(_check_streams, oracle, data, mb, pieces, "soak rotation %d" % rnd)
            members += nm
            nbytes += len(data)
        if pending is not None:
            pending.result()
    finally:
        enc.close()
        pool.shutdown()
    row = {"soak": "members of 64 / 20 / 10 / 10 MiB from %d rotations of the 100 MB workload, -l1, 8 encoders reused, fresh ones every 8th" % rotations,
           "members": members, "bytes": nbytes, "invalid": 0, "seconds": round(time.time() - t0, 1),
           "encode_MBps_8_encoders_host_buffers": round(nbytes / t_enc / 1e6, 1)}
    try:
        with open(os.path.join(ROOT, "gpurun_out", "r06_soak.json"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    print(json.dumps(row))
    assert rotations < 60 or members >= 2000
