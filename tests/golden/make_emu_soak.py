"""Golden hashes for the GPU tier's bar "the GPU's bytes equal the host emulation's at 64 MiB": the members of a soak round
(tools/dev/hunt.py soak_data: 64 MiB members cut from a rotation of the 100 MB workload), each encoded by the HOST EMULATION of the
product's kernels (tests/emu, fast mode, -l1) on the CPU -- about four minutes per member -- and hashed.
    python tests/golden/make_emu_soak.py <round> <member> [<member> ...]   -> merges into tests/golden/emu_soak.json
    python tests/golden/make_emu_soak.py <round> <member> --mib N          -> the member's first N MiB only ("..., first N MiB")
The emulation runs a launch's threads one after another; the GPU tests demand that eight concurrent GPU encoders, fresh and
reused, write these very bytes (tests/test_gpu_soak.py).
The eight encoders of a members job take whole 16 MiB blocks as units; an encoder that has the GPU to itself -- the emulation's
default too -- takes 8 MiB (orz_stream.h, unit_): the goldens are the members job's, hence the ORZ_FAST_UNIT below."""
import ctypes, hashlib, json, os, subprocess, sys, time

os.environ["ORZ_FAST_UNIT"] = str(1 << 24)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tools", "dev")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hunt  # noqa: E402

M = 1 << 26


def emu_lib():
    so = os.path.join(ROOT, "build", "libemu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_backend.cpp", "simt.h")]
    srcs += [os.path.join(ROOT, "orz_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "orz_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    return ctypes.CDLL(so)


def main():
    rnd = int(sys.argv[1])
    args = sys.argv[2:]
    mib = 0
    if "--mib" in args:
        mib = int(args[args.index("--mib") + 1])
        args = args[: args.index("--mib")]
    members = [int(x) for x in args]
    lib = emu_lib()
    data = hunt.soak_data(rnd, 8)
    out = os.path.join(ROOT, "tests", "golden", "emu_soak.json")
    for k in members:
        piece = data[k * M:(k + 1) * M]
        if mib:
            piece = piece[: mib << 20]
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        t0 = time.time()
        rc = lib.emu_encode_fast(piece, ctypes.c_size_t(len(piece)), 15, 9, 6, 0, 0, ctypes.byref(dst), ctypes.byref(n), None)
        assert rc == 0
        blob = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        row = {"bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(), "input_sha256": hashlib.sha256(piece).hexdigest(),
               "emulation_seconds": round(time.time() - t0)}
        # (several of these may run side by side: merge under a lock file)
        import fcntl
        with open(out + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            table = json.load(open(out)) if os.path.exists(out) else {}
            table["round %d member %d" % (rnd, k) + (", first %d MiB" % mib if mib else "")] = row
            with open(out, "w") as f:
                json.dump(table, f, indent=1, sort_keys=True)
        print(k, row, flush=True)


if __name__ == "__main__":
    main()
