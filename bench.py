#!/usr/bin/env python3
"""bench.py -- orz -l1 encode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the full orz encode (`LZEncoder::encode`
driven by `orz::encode`, /root/reference/src/lib.rs:58-92) of the 100,000,000-byte text workload
at -l1, input already resident in HBM, one 16 MiB block in flight (BASELINE configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; every rank encodes its own member
   -- a distinct rotation of the workload -- and the finished bitstreams are gathered to rank 0
   over RCCL; weak scaling.)

Rank 0 prints ONE JSON line.  `roofline` is about the dominant kernel (the speculative parse,
ParseWave): algorithmic bytes per launch = 1.27 B per input byte (read the window
once + write the bitstream, r ~= 0.27) x the input bytes one launch retires on average
(input bytes / launches), over that kernel's average launch duration measured with HIP events on
the encoder's own stream.  `cpu_baseline` times the CPU oracle (a C restatement of the reference
encoder, single thread like the reference) on the same workload on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOAD_BYTES = 100_000_000
LEVEL = 1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_INPUT_BYTE = 1.27


def cpu_baseline(data, level):
    """the oracle (kind "port"), single thread, on this box's host cores; min of 3 full passes"""
    import _oracle

    _oracle.lib()
    best = None
    out_len = 0
    t_all = time.time()
    for _ in range(3):
        t0 = time.time()
        out_len = len(_oracle.encode(data, level))
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
        if time.time() - t_all > 45:
            break
    return {
        "value": round(len(data) / best / 1e6, 2),
        "unit": "MB/s",
        "cores": 1,
        "kind": "port",
        "sample": "full workload (%d bytes), -l%d, min of up to 3 passes, compressed %d bytes" % (len(data), level, out_len),
        "host_cores_available": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=WORKLOAD_BYTES, help="workload size (default: BASELINE config 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import corpus
    import orz_amd
    from orz_amd import dist as odist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # test hook: ORZ_BENCH_BACKEND=gloo + ORZ_BENCH_DEVICE=0 runs several ranks on ONE GPU (RCCL refuses that)
    backend = os.environ.get("ORZ_BENCH_BACKEND", "nccl")
    if "ORZ_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ORZ_BENCH_DEVICE"])
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and distributed:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    base = corpus.text_corpus(args.bytes)
    sha = hashlib.sha256(base).hexdigest()
    # each rank's member is a distinct rotation of the workload (same statistics, different bytes in flight)
    rot = (rank * 12_345_679) % max(1, len(base))
    data = base[rot:] + base[:rot]
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    enc = orz_amd.StreamEncoder(device=local_rank, level=LEVEL)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        out, st = enc.encode_device(src.data_ptr(), src.numel(), stats=True)
        if distributed:  # the job's only exchange: gather the finished bitstreams on rank 0
            got = odist.gather_members({rank: out}, world, rank, world, device=dev if backend == "nccl" else None)
            if rank == 0:
                assert all(g is not None for g in got)
        return out, st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.time()
    agg = {"parse_kernel_ms": 0.0, "parse_launches": 0, "sweeps": 0, "t_prep_s": 0.0, "t_parse_s": 0.0, "t_post_s": 0.0}
    out = b""
    for _ in range(args.steps):
        out, st = step()
        for k in agg:
            agg[k] += st[k]
    barrier()
    dt = time.time() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_bytes = len(data) * world * args.steps
        value = total_bytes / dt / 1e6
        launches = max(1, agg["parse_launches"])
        avg_launch_s = agg["parse_kernel_ms"] / 1e3 / launches
        bytes_per_launch = ALGO_BYTES_PER_INPUT_BYTE * len(data) * args.steps / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE and
        # WRITE_SIZE collected in separate rocprofv3 runs on one 16 MiB block, profiles/): not live
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01g_pmc_hbm_traffic_16MiB_l1.json")) as f:
                traffic = json.load(f).get("parse_wave_hbm_bytes_per_launch")
        except Exception:
            traffic = None
        res = {
            "metric": "orz -l1 encode throughput (enwik8-shaped text, 100 MB, one 16 MiB block in flight)",
            "value": round(value, 3),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic: deterministic text corpus assembled from text files of this image (tools/corpus.py), "
                    "sha256 " + sha[:16],
            "config": {
                "workload": "BASELINE configs[1]: orz -l1, %d bytes of text, single stream per GPU, one 16 MiB block in flight"
                            % len(data),
                "level": LEVEL,
                "lzcfg": [15, 9, 6],
                "members": world,
                "segment_bytes": 62,
                "window_segments": 3072,
                "handoff_lookback_segments": 24,
                "handoff_deadline_us": 110,
                "input": "resident in HBM",
            },
            "compressed_bytes": len(out),
            "ratio": round(len(out) / len(data), 5),
            "roofline": {
                "bound": "hbm",
                "kernel": "orz_wave_kernel<ParseWave>",
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 8),
                "traffic": traffic,
                "launches_per_step": launches // args.steps,
                "avg_launch_us": round(avg_launch_s * 1e6, 2),
                "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
            },
            "stage_seconds_per_step": {
                "prep": round(agg["t_prep_s"] / args.steps, 4),
                "parse": round(agg["t_parse_s"] / args.steps, 4),
                "post": round(agg["t_post_s"] / args.steps, 4),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(base, LEVEL)
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    enc.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
