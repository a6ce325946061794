"""Calibration workload for the FETCH_SIZE counter on this box (MI355X_MICROARCH.md, HBM: "calibrate on a known
byte count in your own access pattern before trusting an absolute").  Two access patterns with known byte
counts, well past the 256 MiB Infinity Cache:
  stream : a 1 GiB tensor read once, wide and coalesced (16 B per lane)        -> the guide's x2 case
  gather : 8 Mi random 32-byte rows out of a 1 GiB table (each row read once)   -> the parse kernel's pattern
           (slot records are 32 B, fetched by scattered 16-byte loads)
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`; tools/pmc_calibrate.py --report <db> prints the counter
per kernel next to the known bytes."""
import json
import sqlite3
import sys

ROWS = 32 * 1024 * 1024  # x 32 B = 1 GiB
PICK = 8 * 1024 * 1024


def workload():
    import torch

    dev = torch.device("cuda", 0)
    table = torch.arange(ROWS * 8, dtype=torch.int32, device=dev).view(ROWS, 8)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    idx = torch.randperm(ROWS, device=dev, generator=g)[:PICK].contiguous()
    torch.cuda.synchronize()
    for _ in range(3):
        s = table.sum(dtype=torch.int64)       # stream: reads ROWS * 32 bytes
        torch.cuda.synchronize()
        out = table.index_select(0, idx)       # gather: reads PICK * 32 useful bytes (+ 8 B index per row)
        torch.cuda.synchronize()
    print(int(s), int(out[0, 0]))


def report(db):
    cur = sqlite3.connect(db).cursor()
    q = """select k.kernel_name, count(*), sum(e.value) from rocpd_pmc_event e
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name"""
    rows = []
    for name, n, total in cur.execute(q):
        kib = total / n
        rows.append({"kernel": name[:90], "dispatches": n, "FETCH_SIZE_KiB_per_dispatch": round(kib, 1)})
    rows.sort(key=lambda r: -r["FETCH_SIZE_KiB_per_dispatch"])
    res = {"known": {"stream_bytes": ROWS * 32, "gather_useful_bytes": PICK * 32, "gather_index_bytes": PICK * 8},
           "kernels": rows[:8]}
    for r in rows:
        b = r["FETCH_SIZE_KiB_per_dispatch"] * 1024
        if "reduce" in r["kernel"].lower() and b > ROWS * 8:
            res["stream_counter_over_known"] = round(b / (ROWS * 32), 3)
        if ("index" in r["kernel"].lower() or "gather" in r["kernel"].lower()) and b > PICK * 8:
            res["gather_counter_over_useful"] = round(b / (PICK * 32), 3)
            res["gather_counter_bytes_per_row"] = round(b / PICK, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        workload()
