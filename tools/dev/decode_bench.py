"""dev: the device decoder's rate per member -- `members` members of `mib` MiB of the text workload encoded on the GPU, decoded by
orz_decode_members_device (one member per wavefront): kernel time -> MB/s per member; the host decoder (one thread) beside it."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import corpus, orz_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
members = int(sys.argv[2]) if len(sys.argv) > 2 else 4
data = corpus.enwik_like(100_000_000)[: members * (mib << 20)]
enc = orz_amd.MemberEncoder(device=0, level=1, jobs=min(members, 8))
blob, n = enc.encode(data, member_bytes=mib << 20)
enc.close()
orz_amd.decode_members_device(blob[:0])
t0 = time.time(); out, m, st = orz_amd.decode_members_device(blob, stats=True); t = time.time() - t0
ok = out == data and m == n
t0 = time.time(); host, _ = orz_amd.decode_members(blob); th = time.time() - t0
print(json.dumps({"members": n, "member_MiB": mib, "exact": bool(ok and host == data), "kernel_ms": round(st["kernel_ms"], 1),
                  "MBps_per_member": round((mib << 20) / (st["kernel_ms"] / 1e3) / 1e6, 2), "device_wall_s": round(t, 2),
                  "host_decoder_MBps_1_thread": round(len(data) / th / 1e6, 1)}))
