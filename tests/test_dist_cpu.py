"""world_size-2 gloo test of the multi-GPU layer's host logic (member sharding + bitstream gather)."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_members, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from orz_amd import dist as od

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # stand-in payloads: member m is a fake "stream" of m-dependent length ending in the EOF byte
        local = {m: bytes([1 + (m * 7 + i) % 250 for i in range(100 + 37 * m)]) for m in od.members_of_rank(n_members, rank, world)}
        got = od.gather_members(local, n_members, rank, world)
        kept = od.gather_members(local, n_members, rank, world, to_host=False)  # members of other ranks stay tensors
        if rank == 0:
            assert [bytes(k) if isinstance(k, bytes) else k.numpy().tobytes() for k in kept] == got
            assert any(not isinstance(k, bytes) for k in kept)
            q.put([len(x) for x in got] + [sum(x[0] for x in got)])
    finally:
        dist.destroy_process_group()


def test_member_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_members = 5
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_members, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = q.get(timeout=10)
    assert res[:-1] == [100 + 37 * m for m in range(n_members)]
    assert res[-1] == sum(1 + (m * 7) % 250 for m in range(n_members))


def test_member_gather_world8():
    """the shape of BASELINE configs[3] -- 8 ranks, 61 members (an 8 GB job in 128 MiB members + a short last one): ranks
    hold 8 or 7 members, rank 0 posts the receives of all seven peers before it waits for any"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_members, world = 61, 8
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_members, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = q.get(timeout=10)
    assert res[:-1] == [100 + 37 * m for m in range(n_members)]
    assert res[-1] == sum(1 + (m * 7) % 250 for m in range(n_members))


def _library_buffer(payload):
    """an api.OrzBuffer like the one orz_stream_encode's output arrives in: malloc'ed memory, released by orz_free"""
    import ctypes

    from orz_amd import _native, api

    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    p = libc.malloc(len(payload))
    ctypes.memmove(p, payload, len(payload))
    return api.OrzBuffer(_native.load(), ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), len(payload))


def _worker_held(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from orz_amd import dist as od

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        payload = _library_buffer(bytes([3 + rank]) * (1000 + 17 * rank))
        got = od.gather_members({rank: payload}, world, rank, world, to_host=False)
        if rank == 0:
            q.put([bytes(g) if not hasattr(g, "numpy") else g.numpy().tobytes() for g in got])
    finally:
        dist.destroy_process_group()


def test_gather_of_library_held_buffers_world2():
    """bench.py's flow: one member per rank, each held in the buffer the library returned (no bytes copy)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_held, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == [bytes([3]) * 1000, bytes([4]) * 1017]


def _worker_real(rank, world, port, n_members, q):
    """every rank encodes ITS members with the oracle (standing in for its GPU), then the real gather"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch.distributed as dist

    import _data
    import _oracle
    from orz_amd import dist as od

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = _data.mixed(300_000, seed=4)
        size = (len(data) + n_members - 1) // n_members
        local = {m: _oracle.encode(data[m * size:(m + 1) * size], 1) for m in od.members_of_rank(n_members, rank, world)}
        got = od.gather_members(local, n_members, rank, world)
        if rank == 0:
            container = b"".join(got)
            pieces = od.split_members(container)
            back = b"".join(_oracle.decode(p)[0] for p in pieces)
            q.put((len(pieces), back == data))
    finally:
        dist.destroy_process_group()


def test_real_streams_gather_and_decode_world2():
    """world-2 job over gloo: members encoded on their ranks, gathered at exact sizes, container cut at the EOF chunks,
    every member decoded by the oracle decoder: the input comes back"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) == (5, True)


def test_split_members_roundtrip(oracle):
    from orz_amd import dist as od

    parts = [oracle.encode(b"member %d " % i * (50 + i), 1) for i in range(4)]
    assert od.split_members(b"".join(parts)) == parts
    for i, p in enumerate(parts):
        assert oracle.decode(p)[0] == b"member %d " % i * (50 + i)


def test_round_robin_assignment():
    from orz_amd import dist as od

    seen = sorted(m for r in range(8) for m in od.members_of_rank(61, r, 8))
    assert seen == list(range(61))
