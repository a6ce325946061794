"""Multi-GPU layer: independent members farmed across ranks, one gather of the finished bitstreams.

A single orz stream does not shard (its model state is one adaptive chain); what
shards is a job of independent *members* (input chunks encoded as complete orz streams).  Member m
goes to rank m % world; each rank encodes its members on its own GPU with no data-path collective;
the only exchange is the final variable-length gather of the compressed members to rank 0
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Container: the members of a job are concatenated in member order.  Each member is a complete orz
stream ending in its own 0x00 EOF chunk, so `split_members` can cut the container again and every
piece decodes with the reference decoder (which reads one stream and stops, src/lib.rs:108-110).
"""
import torch
import torch.distributed as dist


def members_of_rank(n_members, rank, world):
    """member indices encoded by `rank` (round robin)"""
    return list(range(rank, n_members, world))


def gather_members(local, n_members, rank, world, device=None, to_host=True):
    """local: {member index: bytes} encoded on this rank.  Returns the list of all members in member
    order on rank 0 (None elsewhere).  One all-gather of the sizes, then every rank sends exactly its bytes to
    rank 0 (point-to-point: each peer has its own xGMI link to the root, nothing is padded).
    to_host=False leaves the members received from other ranks where they arrived (uint8 tensors on `device`): the
    gather is complete when rank 0 holds the bytes, and a caller that only forwards or sizes them (bench.py) should
    not pay for world-1 device-to-host copies on rank 0."""
    mine = members_of_rank(n_members, rank, world)
    assert sorted(local) == mine
    dev = device if device is not None else torch.device("cpu")
    per_rank = (n_members + world - 1) // world
    sizes = torch.zeros(per_rank, dtype=torch.int64, device=dev)
    for i, m in enumerate(mine):
        sizes[i] = len(local[m])
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    # one payload held in the library's own buffer (api.OrzBuffer: bench.py's case, one member per rank) is sent from
    # where it lies; several members, or plain bytes, are joined first
    from .api import OrzBuffer

    single_buf = len(mine) == 1 and isinstance(local[mine[0]], OrzBuffer)
    # members that lie in device memory already (uint8 tensors on `dev`: StreamEncoder.encode_to_device, round 6) are sent from
    # where they lie -- no copy to the host and back (before: every member went device -> host -> device -> wire)
    on_dev = bool(mine) and all(isinstance(local[m], torch.Tensor) for m in mine)
    if on_dev:
        parts = [local[m].reshape(-1) for m in mine]
        flat = (parts[0] if len(parts) == 1 else torch.cat(parts)) if sum(p.numel() for p in parts) else None
        blob = flat if flat is not None else torch.empty(0, dtype=torch.uint8, device=dev)
        if flat is not None and flat.device != torch.device(dev):
            flat = flat.to(dev)
    elif single_buf:
        blob = local[mine[0]]
        flat = torch.frombuffer(blob.view(), dtype=torch.uint8) if len(blob) else None
    else:
        blob = b"".join(bytes(local[m]) for m in mine)
        flat = torch.frombuffer(bytearray(blob), dtype=torch.uint8) if blob else None
    if rank != 0:
        if flat is not None:
            dist.send(flat.to(dev), dst=0)
        return None
    # rank 0: post the receives of ALL peers first, then wait -- every peer has its own link to the root, so the
    # transfers overlap instead of running in rank order
    bufs, reqs = {}, []
    for r in range(1, world):
        total = int(all_sizes[r].sum())
        if total:
            bufs[r] = torch.empty(total, dtype=torch.uint8, device=dev)
            reqs.append(dist.irecv(bufs[r], src=r))
    for q in reqs:
        q.wait()
    out = [None] * n_members
    for r in range(world):
        ms = members_of_rank(n_members, r, world)
        if r == 0:
            if on_dev:
                raw = blob.cpu().numpy().tobytes() if to_host else blob
            else:
                raw = bytes(blob) if (to_host or not single_buf) else blob  # to_host: always plain bytes
        elif r in bufs:
            raw = bufs[r].cpu().numpy().tobytes() if to_host else bufs[r]
        else:
            raw = b""
        if len(ms) == 1 and not isinstance(raw, (bytes, bytearray)):  # a single payload left where it is (to_host=False)
            out[ms[0]] = raw
            continue
        at = 0
        for i, m in enumerate(ms):
            ln = int(all_sizes[r][i])
            out[m] = raw[at:at + ln]
            at += ln
    return out


def split_members(container):
    """Cut a concatenation of orz streams at their EOF chunks (LEB128 framing, src/ioutil.rs:60-88)."""
    out, at, start = [], 0, 0
    n = len(container)
    while at < n:
        t, shift = 0, 0
        while True:
            b = container[at]
            at += 1
            t |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        if t == 0:
            out.append(container[start:at])
            start = at
        else:
            at += t
    if start != n:
        raise ValueError("container does not end on an EOF chunk")
    return out
