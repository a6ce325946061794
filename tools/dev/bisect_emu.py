"""dev: the GPU's fast-mode stream against the host emulation's for one input, under a knob given in the environment"""
import ctypes, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import corpus, orz_amd
data = corpus.enwik_like(5_000_000)
ref_path = "/tmp/bisect_ref.bin"
if not os.path.exists(ref_path):
    L = ctypes.CDLL(os.path.join(ROOT, "build", "libemu.so"))
    dst = ctypes.POINTER(ctypes.c_uint8)(); n = ctypes.c_size_t()
    assert L.emu_encode_fast(data, ctypes.c_size_t(len(data)), 15, 9, 6, 0, 0, ctypes.byref(dst), ctypes.byref(n), None) == 0
    open(ref_path, "wb").write(ctypes.string_at(dst, n.value))
ref = open(ref_path, "rb").read()
outs = []
for k in range(2):
    enc = orz_amd.StreamEncoder(device=0, level=1, mode="fast")
    outs.append(enc.encode(data)); enc.close()
print(sys.argv[1:], "gpu", [len(o) for o in outs], "same twice:", outs[0] == outs[1], "| emu", len(ref), "| gpu == emu:", outs[0] == ref, flush=True)
