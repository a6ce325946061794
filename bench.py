#!/usr/bin/env python3
"""bench.py -- orz -l1 encode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the full orz encode (`LZEncoder::encode`
driven by `orz::encode`, /root/reference/src/lib.rs:58-92) of the 100,000,000-byte enwik8-shaped
workload (tools/enwik_like.py: seeded, box-independent, SHA-256 pinned) at -l1, input already
resident in HBM, one stream, one 16 MiB block in flight (BASELINE configs[1]).

  python bench.py --gpus N --steps K --warmup W [--mode fast|exact]
  (N > 1: launched by torch.distributed.run, one rank per GPU; every rank encodes its own member
   -- a distinct rotation of the workload -- and the finished bitstreams are gathered to rank 0
   over RCCL; weak scaling.)

Rank 0 prints ONE JSON line.  `config` is read back from the encoder (orz_stream_get_config), not
written down here.  `roofline` describes the kernel with the largest share of device time, from HIP
events recorded inside the library on the stream each kernel runs on (orz_stream_get_kernel_times);
`roofline_others` carries the next ones.  Algorithmic bytes per launch = 1.27 B per input byte
(read the window once + write the bitstream, r ~= 0.27) x the input bytes one launch
retires (input bytes / launches).  `traffic` is the PMC-measured HBM bytes per launch from this
round's rocprofv3 passes (profiles/, FETCH_SIZE and WRITE_SIZE collected separately) when that file
names the same kernel, else null.  `cpu_baseline` times the CPU oracle (a C restatement of the
reference encoder, single thread like the reference) on the same workload on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import sys
import time

# an encoder drives three HIP streams; the runtime's default of 4 hardware queues makes streams share queues (the library
# sets the same default when it is loaded first -- here torch initialises HIP before it)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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
        "compressed_bytes": out_len,
    }


def _meminfo_available_bytes():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return None


def members_leg(base, level, jobs=8, member_bytes=1 << 26, total=16 << 26, device=0, nproc=None):
    """Outside the timed region, rank 0 at N=1 only: the aggregate path of BASELINE configs[2]/[3] on ONE GPU and its CPU baseline
    (the reference's harness, /root/reference/benchmark-tool/src/main.rs:57-114, times an encoder
    process and verifies its output by decoding it).
      members               `jobs` stream encoders on the GPU over `total` bytes of the workload cut into members of
                            `member_bytes` (input resident in HBM) -> MB/s; every member's stream through the ORACLE's decoder
                            (one `oracle/orz_oracle decode` process each); size against the oracle's encoder on the same split
      cpu_baseline_members  one `oracle/orz_oracle encode` process per host core, each encoding one of the same members, all
                            started together: bytes encoded / wall time (tmpfs file reads and writes included)
    """
    import torch

    import orz_amd
    from orz_amd import dist as odist

    data = (base * (total // len(base) + 1))[:total]
    members = [data[i:i + member_bytes] for i in range(0, len(data), member_bytes)]
    # ---- GPU: `jobs` encoders, input in HBM
    enc = orz_amd.MemberEncoder(device=device, level=level, jobs=jobs)
    enc.encode(data[: jobs * (1 << 20)], member_bytes=1 << 20)  # warm-up: allocations, first launches
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(torch.device("cuda", device))
    torch.cuda.synchronize()
    # The job runs twice and the faster pass counts (as the CPU baseline takes the best of its passes): the encoders' per-item
    # buffers and the job's device arena grow the first time they see a job of this size; a service reuses its encoders
    t_gpu = None
    for _ in range(2):
        t0 = time.time()
        blob, n = enc.encode_device(src.data_ptr(), src.numel(), member_bytes=member_bytes)
        dt = time.time() - t0
        t_gpu = dt if t_gpu is None else min(t_gpu, dt)
    enc.close()
    del src
    streams = odist.split_members(blob)
    assert n == len(members) == len(streams)
    res = members_check_and_cpu(members, streams, level, nproc=nproc)
    gpu_mbs = len(data) / t_gpu / 1e6
    res["members"].update({"value": round(gpu_mbs, 1), "unit": "MB/s", "encoders_on_one_gpu": jobs, "seconds": round(t_gpu, 3), "passes": "faster of two",
                           "input": "resident in HBM",
                           "pipeline_frac_of_hbm_peak": round(ALGO_BYTES_PER_INPUT_BYTE * gpu_mbs / 1e3 / HBM_PEAK_GBS, 8)})
    res["gpu_over_cpu_members"] = round(gpu_mbs / res["cpu_baseline_members"]["value"], 4)
    return res


def members_check_and_cpu(members, streams, level, max_procs=None, nproc=None):
    """the host half of members_leg: `streams[k]` must decode to `members[k]` with the oracle's decoder (a process each);
    the oracle's encoder on the same members, one process per host core, timed"""
    import shutil
    import subprocess
    import tempfile

    import _oracle

    _oracle.lib()  # (builds oracle/ when absent)
    cli = _oracle.CLI
    n = len(members)
    total = sum(len(m) for m in members)
    member_bytes = len(members[0]) if members else 0
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 12 * total // 10 + (8 << 30) else None
    work = tempfile.mkdtemp(prefix="orz_bench_", dir=root)
    try:
        for k, m in enumerate(members):
            with open(os.path.join(work, "m%d.in" % k), "wb") as f:
                f.write(m)
            with open(os.path.join(work, "g%d.orz" % k), "wb") as f:
                f.write(streams[k])
        # ---- every member's stream through the oracle's decoder
        t0 = time.time()
        procs = [subprocess.Popen([cli, "decode", os.path.join(work, "g%d.orz" % k), os.path.join(work, "g%d.out" % k)],
                                  stderr=subprocess.DEVNULL) for k in range(n)]
        rcs = [p.wait() for p in procs]
        bad = []
        for k in range(n):
            ok = rcs[k] == 0
            if ok:
                with open(os.path.join(work, "g%d.out" % k), "rb") as f:
                    ok = hashlib.sha256(f.read()).digest() == hashlib.sha256(members[k]).digest()
            if not ok:
                bad.append(k)
            try:
                os.remove(os.path.join(work, "g%d.out" % k))
            except OSError:
                pass
        t_dec = time.time() - t0
        # ---- CPU: one oracle process per host core.  How many of the cores the box reports actually run side by side is
        # measured first (a ladder of process counts on an 8 MiB slice: VMs report more cores than they deliver), the member
        # pass then uses the count with the best aggregate, bounded by memory (a process holds ~0.25 GB at 64 MiB members)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        avail = _meminfo_available_bytes()
        cap = cores if avail is None else max(1, min(cores, int(avail * 0.6) // (256 << 20)))
        if max_procs:
            cap = min(cap, max_procs)
        slice_bytes = min(8 << 20, member_bytes)
        with open(os.path.join(work, "slice.in"), "wb") as f:
            f.write(members[0][:slice_bytes])

        def run_encoders(count, src_of, dst_of):
            t0 = time.time()
            ps = [subprocess.Popen([cli, "encode", "-l%d" % level, src_of(k), dst_of(k)], stderr=subprocess.DEVNULL) for k in range(count)]
            ok = all(p.wait() == 0 for p in ps)
            assert ok, "an oracle encoder process failed"
            return time.time() - t0

        ladder, c = [], 1
        while c < cap:
            ladder.append(c)
            c *= 4
        ladder.append(cap)
        if nproc:  # (a later leg of the same run: the process count the first leg's ladder found best, probed once at this level)
            ladder = [min(nproc, cap)]
        probe = []
        for c in ladder:
            dt = run_encoders(c, lambda k: os.path.join(work, "slice.in"), lambda k: os.path.join(work, "s%d.orz" % k))
            probe.append({"processes": c, "MBps": round(c * slice_bytes / dt / 1e6, 1), "seconds": round(dt, 2)})
            if dt > 20:  # (more processes only take longer from here)
                break
        best = max(probe, key=lambda r: r["MBps"])
        nproc = best["processes"]
        predicted = nproc * member_bytes / (best["MBps"] * 1e6)
        if predicted <= 40:
            t_cpu = run_encoders(nproc, lambda k: os.path.join(work, "m%d.in" % (k % n)), lambda k: os.path.join(work, "c%d.orz" % k))
            cpu_bytes = sum(len(members[k % n]) for k in range(nproc))
            sample = ("%d oracle processes started together (the count with the best aggregate of the ladder in `scaling_probe`; the box "
                      "reports %d cores), each encoding one %d-byte member of the same split (-l%d); %d bytes in %.2f s wall, file reads "
                      "and writes on %s included" % (nproc, cores, member_bytes, level, cpu_bytes, t_cpu, "tmpfs" if root else "the temp dir"))
        else:  # the member pass would not fit the bench's time budget on this host: the probe's figure stands
            t_cpu, cpu_bytes = best["seconds"], nproc * slice_bytes
            sample = ("%d oracle processes started together, each encoding the first %d bytes of member 0 (-l%d): the full member pass was "
                      "predicted at %.0f s on this host and skipped" % (nproc, slice_bytes, level, predicted))
        # sizes of the same split by the oracle's encoder (members the per-core pass did not reach are encoded now)
        missing = [k for k in range(n) if k >= nproc or predicted > 40]
        for lo in range(0, len(missing), nproc):
            ps = [subprocess.Popen([cli, "encode", "-l%d" % level, os.path.join(work, "m%d.in" % k), os.path.join(work, "c%d.orz" % k)])
                  for k in missing[lo:lo + nproc]]
            assert all(p.wait() == 0 for p in ps)
        ref_sizes = [os.path.getsize(os.path.join(work, "c%d.orz" % k)) for k in range(n)]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    gpu_sizes = [len(s) for s in streams]
    cpu_mbs = cpu_bytes / t_cpu / 1e6
    return {
        "members": {
            "members": n, "member_bytes": member_bytes, "bytes": total, "level": level,
            "compressed_bytes": sum(gpu_sizes), "oracle_compressed_bytes_same_split": sum(ref_sizes),
            "size_delta_pct": round(100.0 * (sum(gpu_sizes) - sum(ref_sizes)) / max(1, sum(ref_sizes)), 4),
            "roundtrip_ok": not bad, "members_not_decoding": bad,
            "roundtrip_checker": "oracle decoder, one process per member, %.2f s" % t_dec,
        },
        "cpu_baseline_members": {
            "value": round(cpu_mbs, 1), "unit": "MB/s", "cores": nproc, "kind": "port", "host_cores_available": cores,
            "sample": sample, "per_core_MBps": round(cpu_mbs / nproc, 2), "scaling_probe": probe,
        },
    }


def kernel_table_rows(ktable, nbytes, pmc):
    blocks = max(1.0, nbytes / float(1 << 24))
    rows = []
    for name, ms, n in ktable:
        if not n:
            continue
        tr = None
        base = name.split("<")[0].split(" ")[0]
        for k, v in (pmc or {}).items():
            if k.endswith("<" + base + ">") or k == name or (base == "orz_symrank_kernel" and k == base):
                tr = v
        rows.append({"kernel": name, "launches_per_block": round(n / blocks, 1), "avg_launch_us": round(ms / n * 1e3, 2),
                     "ms_per_block": round(ms / blocks, 3), "hbm_bytes_per_launch": tr,
                     "hbm_gbs": round(tr / (ms / n / 1e3) / 1e9, 1) if tr else None})
    total = sum(r["ms_per_block"] for r in rows if "symrank" not in r["kernel"])
    return {"per": "16 MiB block (workload bytes / 2^24 blocks)", "sum_ms_per_block_without_symbol_ranking": round(total, 2), "rows": rows}


def oracle_check(data, stream):
    """outside the timed region: the stream the timed passes produced goes through the ORACLE's decoder"""
    import _oracle

    _oracle.lib()
    t0 = time.time()
    try:
        back, used = _oracle.decode(stream)
        ok = used == len(stream) and back == data
    except Exception:
        ok = False
    return ok, round(time.time() - t0, 2)


def newest_pmc_profile():
    """HBM bytes per launch from the builder's PMC passes (tools/profile_round.sh; FETCH_SIZE and WRITE_SIZE collected in
    separate rocprofv3 runs): the newest profiles/rNN_pmc_hbm_traffic_16MiB_l1.json"""
    d = os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(d) if n.endswith("_pmc_hbm_traffic_16MiB_l1.json"))
    except OSError:
        return None, {}
    for name in reversed(names):
        try:
            with open(os.path.join(d, name)) as f:
                return "profiles/" + name, json.load(f).get("by_name", {})
        except Exception:
            continue
    return None, {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=WORKLOAD_BYTES, help="workload size (default: BASELINE config 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-members", action="store_true", help="skip the members legs (--members-jobs encoders on one GPU + their many-core CPU baselines)")
    # (round 6: eight encoders as before.  Twelve were measured -- 783..790 against 710..746 MB/s in one call, 723..740 against 770
    # in another, sixteen 731: no difference that survives a change of box, profiles/r06_members_jobs.jsonl -- and sixteen members,
    # 1 GiB, so that both halves of the job keep every encoder busy: until now 1e9 bytes = fourteen members and a part)
    ap.add_argument("--members-jobs", type=int, default=8, help="stream encoders on the GPU in the members legs")
    ap.add_argument("--members-bytes", type=int, default=16 << 26, help="bytes of the -l1 members leg (64 MiB members)")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast",
                    help="fast: GPU-native parse (reference-decodable, size within +-0.5 %%); exact: the reference's parse item for item")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import corpus
    import orz_amd
    from orz_amd import dist as odist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (ORZ_BENCH_FORCE_DIST=1: the distributed branch at world size 1 -- tests/test_gpu_device_output.py runs it under torchrun with
    # the nccl backend so that the RCCL path has executed on the one GPU a test box has)
    distributed = world > 1 or (os.environ.get("ORZ_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
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

    base = corpus.enwik_like(args.bytes)  # raises if the canonical 100,000,000 bytes do not hash to the pinned SHA-256
    sha = hashlib.sha256(base).hexdigest()
    if args.bytes == WORKLOAD_BYTES:
        import enwik_like

        assert sha == enwik_like.MANIFEST[WORKLOAD_BYTES], "workload differs from the pinned one"
    # each rank's member is a distinct rotation of the workload (same statistics, different bytes in flight)
    rot = (rank * 12_345_679) % max(1, len(base))
    data = base[rot:] + base[:rot]
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    enc = orz_amd.StreamEncoder(device=local_rank, level=LEVEL, mode=args.mode)
    cfg = enc.config()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the finished stream stays in HBM (orz_stream_encode_to_device, framed on the device) and the gather sends it from
    # there; N = 1: the headline's timed region ends with the stream in a host buffer (orz_stream_encode), as before
    dout = torch.empty(orz_amd.stream_bound(src.numel()), dtype=torch.uint8, device=dev) if distributed else None

    def step():
        if distributed:
            n_out, st = enc.encode_to_device(src.data_ptr(), src.numel(), dout.data_ptr(), dout.numel(), stats=True)
            out = dout[:n_out]
        else:
            out, st = enc.encode_device(src.data_ptr(), src.numel(), stats=True, raw=True)
        if distributed:  # the job's only exchange: gather the finished bitstreams on rank 0
            tg = time.time()
            payload = out if backend == "nccl" else out.cpu()
            got = odist.gather_members({rank: payload}, world, rank, world, device=dev if backend == "nccl" else None, to_host=False)
            if rank == 0:  # (the members of the other ranks stay in rank 0's HBM: gathered, not copied out again)
                assert all(g is not None and len(g) > 0 for g in got)
            gather_s[0] += time.time() - tg
        return out, st

    gather_s = [0.0]
    for _ in range(args.warmup):
        step()
    barrier()
    gather_s[0] = 0.0
    t0 = time.time()
    agg = {"sweeps": 0, "seg_evals": 0, "items": 0, "t_prep_s": 0.0, "t_parse_s": 0.0, "t_post_s": 0.0, "host_syncs": 0, "blocks": 0}
    kt = [[0.0, 0] for _ in range(4)]
    out = b""
    for _ in range(args.steps):
        out, st = step()
        for k in agg:
            agg[k] += st[k]
    dt_own = time.time() - t0  # this rank's own clock up to its last stream (before the closing barrier)
    barrier()
    dt = time.time() - t0
    gather_ms = gather_s[0] / max(1, args.steps) * 1e3
    # roofline leg, outside the timed region: one more pass in profile mode (HIP-event brackets around the kernels of
    # the round loop, which the timed passes replay as a hipGraph), scaled to `steps` passes below
    enc.set_profile(True)
    step()
    for i, (ms, n) in enumerate(enc.kernel_times()):
        kt[i][0] = ms * args.steps
        kt[i][1] = n * args.steps
    ktable = enc.kernel_table()  # every kernel of the profiled pass by name
    enc.set_profile(False)
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank figures for reading the scaling line: each rank's own MB/s and its share of the gather
        mine = torch.tensor([len(data) * args.steps / dt_own / 1e6, gather_ms], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(x[0]), 1), round(float(x[1]), 2)] for x in allr]
    else:
        per_rank = [[round(len(data) * args.steps / dt_own / 1e6, 1), 0.0]]

    if rank == 0:
        total_bytes = len(data) * world * args.steps
        value = total_bytes / dt / 1e6
        names = ["orz_wave_kernel<ParseWave>" if args.mode == "exact" else "orz_thread_kernel<FastEval>", "orz_symrank_kernel",
                 "orz_wave_kernel<FastRowsWave>", "orz_wave_kernel<PathUpWave>"]
        # HBM bytes per launch from the builder's PMC passes (NOT this run: rocprofv3 counters need their own passes), only if
        # the file names the same kernel; labelled as such in `traffic_source`
        pmc_file, pmc = newest_pmc_profile()

        def roof(i):
            ms, n = kt[i]
            if not n or ms <= 0:
                return None
            avg_s = ms / 1e3 / n
            bpl = ALGO_BYTES_PER_INPUT_BYTE * len(data) * args.steps / n
            ach = bpl / avg_s / 1e9
            tr = pmc.get(names[i])
            return {"bound": "hbm", "kernel": names[i], "achieved": round(ach, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 8), "traffic": tr,
                    "traffic_source": ("from_profile: " + pmc_file) if tr is not None else None,
                    # measured HBM rate of this kernel: counter bytes per launch / average launch duration of THIS run
                    "hbm_gbs": round(tr / avg_s / 1e9, 3) if tr is not None else None,
                    "hbm_frac_of_peak": round(tr / avg_s / 1e9 / HBM_PEAK_GBS, 6) if tr is not None else None,
                    "launches_per_step": n // args.steps,
                    "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": round(bpl, 1),
                    "device_ms_per_step": round(ms / args.steps, 2)}

        order = sorted(range(4), key=lambda i: -kt[i][0])
        roofs = [r for r in (roof(i) for i in order) if r]
        # ... and every other kernel of the profiled pass that takes a millisecond or more of a 16 MiB block (task 7):
        # the same arithmetic from the per-name table (one profiled pass over the workload)
        blocks = max(1.0, len(data) / float(1 << 24))
        seen_k = {r["kernel"].split("<")[-1].rstrip(">") for r in roofs}
        for name, ms, n in ktable:
            kname = name.split("<")[0].split(" ")[0]
            if not n or kname in seen_k or "symrank" in kname or ms / blocks < 1.0:
                continue
            avg_s = ms / 1e3 / n
            bpl = ALGO_BYTES_PER_INPUT_BYTE * len(data) / n
            tr = None
            for k, v in (pmc or {}).items():
                if k.endswith("<" + kname + ">"):
                    tr = v
            roofs.append({"bound": "hbm", "kernel": name, "achieved": round(bpl / avg_s / 1e9, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(bpl / avg_s / 1e9 / HBM_PEAK_GBS, 8), "traffic": tr,
                          "traffic_source": ("from_profile: " + pmc_file) if tr is not None else None,
                          "hbm_gbs": round(tr / avg_s / 1e9, 3) if tr is not None else None,
                          "hbm_frac_of_peak": round(tr / avg_s / 1e9 / HBM_PEAK_GBS, 6) if tr is not None else None,
                          "launches_per_step": n, "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": round(bpl, 1),
                          "device_ms_per_step": round(ms, 2)})
        out_bytes = out.cpu().numpy().tobytes() if isinstance(out, torch.Tensor) else bytes(out)
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
            "data": "synthetic: enwik8-shaped text from a committed word-level Markov model, seeded (tools/enwik_like.py; gzip -6 "
                    "36.4 %, bzip2 -9 29.1 % like enwik8), sha256 " + sha[:16],
            "config": {
                "workload": "BASELINE configs[1]: orz -l1, %d bytes of enwik8-shaped text, single stream per GPU, one 16 MiB block in flight"
                            % len(data) + ("" if not cfg.get("unit_bytes") or cfg["unit_bytes"] >= (1 << 24) else
                                           " (parsed in units of %d bytes: the ranking of a unit beside the parse of the next)" % cfg["unit_bytes"]),
                "mode": "fast" if cfg["mode"] == 1 else "exact",
                "level": LEVEL,
                "lzcfg": [int(enc.cfg.match_depth), int(enc.cfg.lazy_match_depth1), int(enc.cfg.lazy_match_depth2)],
                "members": world,
                "encoder": cfg,  # read back through orz_stream_get_config
                "input": "resident in HBM",
            },
            "compressed_bytes": len(out_bytes),
            "compressed_sha256": hashlib.sha256(out_bytes).hexdigest(),  # (profiles/r05_emulation_100MB.txt: the host emulation's stream of the same workload)
            "ratio": round(len(out_bytes) / len(data), 5),
            "items_per_byte": round(agg["items"] / args.steps / len(data), 4),
            "rounds_or_sweeps_per_step": agg["sweeps"] // args.steps,
            "repairs_per_step": agg["seg_evals"] // args.steps if cfg["mode"] == 1 else None,
            # times the host waited for a stream, per 16 MiB block (orz_encode_stats.host_syncs: one for the block's control block +
            # item count, two when its output is collected two blocks later -- sizes, then bytes --, the closing waits of the call)
            "host_syncs_per_block": round(agg["host_syncs"] / max(1, args.steps * -(-len(data) // (1 << 24))), 2),
            # (a stream of its own parses a block as units -- config.encoder.unit_bytes, orz_stream.h -- and the wait is the unit's)
            "host_syncs_per_unit": round(agg["host_syncs"] / max(1, args.steps * -(-len(data) // max(1, cfg.get("unit_bytes") or (1 << 24)))), 2),
            "roofline": roofs[0] if roofs else None,
            "roofline_others": roofs[1:],
            # every kernel of one profiled pass over the workload (HIP events around each launch, on the stream it runs on; no graph
            # replay in that pass), per 16 MiB block: launches, average launch, ms; HBM bytes per launch where the PMC file has them
            "kernel_table": kernel_table_rows(ktable, len(data), pmc),
            # the whole pipeline against the same roofline: algorithmic bytes of a pass / wall time of a pass
            "pipeline": {"bound": "hbm", "achieved": round(ALGO_BYTES_PER_INPUT_BYTE * len(data) * world / (dt / args.steps) / 1e9, 4),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ALGO_BYTES_PER_INPUT_BYTE * len(data) * world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 8),
                         "algorithmic_bytes_per_step": int(ALGO_BYTES_PER_INPUT_BYTE * len(data) * world)},
            "world_size": dist.get_world_size() if distributed else 1,
            "backend": (backend if distributed else None),
            "per_rank_MBps": [r[0] for r in per_rank],
            "gather_ms_per_step": [r[1] for r in per_rank],
            "stage_seconds_per_step": {
                "prep": round(agg["t_prep_s"] / args.steps, 4),
                "parse": round(agg["t_parse_s"] / args.steps, 4),
                "post": round(agg["t_post_s"] / args.steps, 4),
            },
        }
        # parity of what was timed, outside the timed region: the last timed stream through the oracle's decoder, and its
        # size against the oracle's encoder (the cpu_baseline leg encodes the same bytes)
        ok, dec_s = oracle_check(data, out_bytes)
        res["roundtrip_ok"] = bool(ok)
        res["roundtrip_checker"] = "oracle decoder (oracle/orz_oracle.c), %.2f s" % dec_s
        res["size_delta_pct"] = None
        if not args.no_cpu_baseline:  # (rank 0, whatever N: the same one-thread oracle pass over the unrotated workload)
            res["cpu_baseline"] = cpu_baseline(base, LEVEL)
            ref = res["cpu_baseline"]["compressed_bytes"]
            res["size_delta_pct"] = round(100.0 * (len(out_bytes) - ref) / ref, 4)
        if world == 1 and not args.no_members:
            enc.close()  # (the lone encoder's ~5 GB go back before the members' encoders are built)
            try:
                res.update(members_leg(base, LEVEL, jobs=args.members_jobs, total=args.members_bytes, device=local_rank))
            except Exception as e:  # the headline line must not be lost to its annex
                res["members"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # BASELINE configs[2] / configs[4] at -l2 (src/main.rs:97-102: 45/27/18), a 64 MiB member per encoder each, every member
            # through the oracle's decoder, size against the oracle on the same split, the many-core CPU baseline at -l2 beside it
            nproc = (res.get("cpu_baseline_members") or {}).get("cores")
            l2_total = min(args.members_bytes, args.members_jobs << 26)
            for key, make in (("members_l2_text", lambda: base), ("members_l2_zeros", lambda: corpus.zeros_noise(l2_total))):
                try:
                    leg = members_leg(make(), 2, jobs=args.members_jobs, total=l2_total, device=local_rank, nproc=nproc)
                    leg["members"]["cpu_baseline_members"] = leg["cpu_baseline_members"]
                    leg["members"]["gpu_over_cpu_members"] = leg["gpu_over_cpu_members"]
                    leg["members"]["workload"] = ("BASELINE configs[2]: enwik8-shaped text, -l2" if key.endswith("text") else
                                                  "BASELINE configs[4]: zeros + 1 % noise, -l2") + ", %d bytes in 64 MiB members" % l2_total
                    res[key] = leg["members"]
                except Exception as e:
                    res[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(res), flush=True)
    enc.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
