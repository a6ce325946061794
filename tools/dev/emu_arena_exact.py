"""dev: the exact mode on the host emulation with the encoder's buffers side by side (ORZ_EMU_ARENA_MB=7000 python tools/dev/emu_arena_exact.py [bytes]):
the input of the object-level test whose full first block exposed rebuild_summaries' out-of-bounds write in round 4 (57 minutes of CPU at full size)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT+'/tests'); sys.path.insert(0,ROOT+'/tools')
import _data, _oracle
lib=ctypes.CDLL('' + os.environ.get('ORZ_EMU_LIB', os.path.join(ROOT, 'build', 'libemu.so')) + '')
data = _data.random_bytes(1_150_000) + _data.mixed(16_777_216 - 1_150_000 + 500_000, seed=61)
n=int(sys.argv[1]) if len(sys.argv)>1 else len(data)
data=data[:n]
dst=ctypes.POINTER(ctypes.c_uint8)(); m=ctypes.c_size_t(); st=(ctypes.c_ulonglong*5)()
t=time.time()
rc=lib.emu_encode(bytes(data), ctypes.c_size_t(len(data)), 5,3,2, 62, 256, 1, ctypes.byref(dst), ctypes.byref(m), st)
print('rc',rc,'bytes',m.value,'secs',round(time.time()-t), flush=True)
if rc==0:
    out=ctypes.string_at(dst,m.value)
    print('equals oracle', out==_oracle.encode(data,0))
