"""GPU tier: the finished stream left in DEVICE memory (round 6) -- orz_stream_encode_to_device / orz_members_encode_to_device frame
{ LEB128(t) chunk[t] }* and the EOF byte ON the device (/root/reference/src/lib.rs:79-80,89; src/ioutil.rs:79-88) into a buffer the
caller owns, the analogue of the reference's caller-owned `tbuf` (src/lz.rs:89-95).  Bars: the same bytes as the host-result entry
points write (which the other files hold against the oracle), every stream through the ORACLE's decoder, a buffer that is too small
fails the encode and leaves the encoder usable, and the RCCL gather of bench.py runs once at world size 1."""
import os
import subprocess
import sys

import pytest

import _data

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _dev(data):
    import torch

    return torch.frombuffer(bytearray(data) if data else bytearray(1), dtype=torch.uint8).to("cuda:0")


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("shape", ["empty", "one", "text", "mixed", "two_blocks"])
def test_encode_to_device_writes_what_encode_writes(oracle, shape, mode):
    import corpus
    import torch

    import orz_amd

    if mode == "exact" and shape == "two_blocks":
        pytest.skip("the exact mode's two-block case lives in test_gpu_parity.py")
    data = {"empty": lambda: b"", "one": lambda: b"a", "text": lambda: _data.text(700_000, seed=9), "mixed": lambda: _data.mixed(500_000, seed=4),
            "two_blocks": lambda: corpus.enwik_like(40_000_000)[: (1 << 24) + 3_000_001]}[shape]()
    enc = orz_amd.StreamEncoder(device=0, level=1, mode=mode)
    try:
        want = enc.encode(data)
        src = _dev(data)
        cap = orz_amd.stream_bound(len(data))
        dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda:0")
        dst[cap:] = 0x5A  # (canary behind the buffer)
        n, st = enc.encode_to_device(src.data_ptr(), len(data), dst.data_ptr(), cap, stats=True)
        got = dst[:n].cpu().numpy().tobytes()
        assert got == want
        assert bytes(dst[cap:].cpu().numpy()) == b"\x5a" * 64
        back, used = oracle.decode(got)
        assert used == len(got) and back == data
        if len(data) > (1 << 24):  # two waits per unit for the parse (its read-back; the history count behind a slide by less than a
            # block: a stream of its own parses 8 MiB units), one for the stream (+ the pageable upload's none: input in HBM)
            assert st["host_syncs"] <= 2 * st["blocks"] + 2, st
    finally:
        enc.close()


def test_a_buffer_that_is_too_small_fails_the_encode_and_the_encoder_lives_on(oracle):
    import torch

    import orz_amd

    data = _data.text(900_000, seed=3)
    enc = orz_amd.StreamEncoder(device=0, level=1)
    try:
        want = enc.encode(data)
        src = _dev(data)
        small = torch.zeros(len(want) // 2 + 64, dtype=torch.uint8, device="cuda:0")
        small[len(want) // 2:] = 0x5A
        with pytest.raises(orz_amd.OrzError, match="too small"):
            enc.encode_to_device(src.data_ptr(), len(data), small.data_ptr(), len(want) // 2)
        assert bytes(small[len(want) // 2:].cpu().numpy()) == b"\x5a" * 64  # nothing written past the capacity
        exact = torch.zeros(len(want), dtype=torch.uint8, device="cuda:0")
        n = enc.encode_to_device(src.data_ptr(), len(data), exact.data_ptr(), len(want))  # a buffer of exactly the stream's size is enough
        assert n == len(want) and exact.cpu().numpy().tobytes() == want
    finally:
        enc.close()


def test_gate_finding_leaves_nothing_of_the_block_in_the_device_buffer(monkeypatch):
    """the device frames a block only when its guards have no finding: with a defect injected behind the parse the encode fails
    and the buffer holds no byte of the stream (tests/test_gpu_verify.py holds the classes; this is the device-output side)"""
    import torch

    import orz_amd

    data = _data.text(400_000, seed=8)
    monkeypatch.setenv("ORZ_VERIFY_INJECT", "hole:100")
    enc = orz_amd.StreamEncoder(device=0, level=1)
    try:
        src = _dev(data)
        cap = orz_amd.stream_bound(len(data))
        dst = torch.full((cap,), 0x5A, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(orz_amd.OrzError, match="validity gate"):
            enc.encode_to_device(src.data_ptr(), len(data), dst.data_ptr(), cap)
        assert int((dst != 0x5A).sum()) == 0
    finally:
        enc.close()


def test_members_to_device_are_the_members(oracle):
    import torch

    import orz_amd
    from orz_amd import dist as odist

    data = _data.text(5_300_000, seed=12)
    mb = 1 << 20
    enc = orz_amd.MemberEncoder(device=0, level=1, jobs=3)
    try:
        blob, n = enc.encode(data, member_bytes=mb)
        want = odist.split_members(blob)
        src = _dev(data)
        cap = orz_amd.stream_bound(mb) * n
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        places = enc.encode_to_device(src.data_ptr(), len(data), dst.data_ptr(), cap, member_bytes=mb)
        assert len(places) == n == len(want)
        host = dst.cpu().numpy().tobytes()
        spans = sorted(places)
        for (o0, l0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + l0 <= o1  # members do not overlap in the arena
        for k, (off, ln) in enumerate(places):
            assert host[off:off + ln] == want[k]
            back, used = oracle.decode(want[k])
            assert back == data[k * mb:(k + 1) * mb] and used == ln
        with pytest.raises(orz_amd.OrzError, match="too small"):
            enc.encode_to_device(src.data_ptr(), len(data), dst.data_ptr(), 100_000, member_bytes=mb)
        blob2, _ = enc.encode(data, member_bytes=mb)  # (the workers live on)
        assert blob2 == blob
    finally:
        enc.close()


def test_bench_runs_under_torchrun_with_the_rccl_backend_at_world_size_one(tmp_path):
    """`bench.py --gpus 1` launched the way the driver launches N > 1 -- torch.distributed.run, backend nccl (= RCCL) -- so that
    the distributed branch (process group, sizes all-gather, the rank-0 path of the gather, the stream encoded into HBM and sent
    from there) has executed once on real hardware before a driver gives it eight GPUs"""
    import json

    env = dict(os.environ, ORZ_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--bytes", "40000000",
           "--no-members", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["backend"] == "nccl" and res["world_size"] == 1
    assert res["roundtrip_ok"] is True
    assert res["value"] > 50
