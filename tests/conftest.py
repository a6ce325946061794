import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# GPU tier order: the per-row parity tests first, the full-size configurations and the soak last -- the tier runs with -x, and a
# late failure in the heaviest, newest tests must not blank the per-row evidence ("what's weak" 2)
_GPU_ORDER = ["test_gpu_parity.py", "test_gpu_fast.py", "test_gpu_device_output.py", "test_device_decoder.py", "test_benchmark_tool.py", "test_enwik8.py",
              "test_gpu_verify.py", "test_gpu_arena.py", "test_gpu_configs.py", "test_gpu_soak.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else -1

    items.sort(key=rank)  # (stable: the order inside a file, and of the CPU tests, stays as collected)


def _have_gpu():
    try:
        from orz_amd import _native

        return _native.load().orz_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import _oracle

    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def emu():
    """Host emulation of the device pipeline (tests/emu): same kernel bodies, CPU loops."""
    import ctypes

    so = os.path.join(ROOT, "build", "libemu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_backend.cpp", "simt.h")]
    srcs += [os.path.join(ROOT, "orz_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "orz_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    lib = ctypes.CDLL(so)

    def encode(data, cfg=(15, 9, 6), seg=62, win=256, order=1):
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        st = (ctypes.c_ulonglong * 5)()
        data = bytes(data)
        rc = lib.emu_encode(data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], seg, win, order, ctypes.byref(dst),
                            ctypes.byref(n), st)
        assert rc == 0
        out = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        return out, list(st)

    def encode_fast(data, cfg=(15, 9, 6), tile=0, rounds=0):
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        st = (ctypes.c_ulonglong * 5)()
        data = bytes(data)
        rc = lib.emu_encode_fast(data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], tile, rounds, ctypes.byref(dst),
                                 ctypes.byref(n), st)
        assert rc == 0
        out = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        return out, list(st)

    def encode_fast_reused(first, data, cfg=(15, 9, 6)):
        """`data` through an encoder that encoded `first` before: the second stream"""
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        first, data = bytes(first), bytes(data)
        rc = lib.emu_encode_fast_reused(first, ctypes.c_size_t(len(first)), data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2],
                                        ctypes.byref(dst), ctypes.byref(n))
        assert rc == 0
        out = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        return out

    def encode_fast_device(data, cfg=(15, 9, 6), cap=0):
        """the fast mode through the device-output path (the stream framed by FrameChunks / FrameAdvance / FrameEof): the stream,
        or (None, message) when the encode fails"""
        dst = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        data = bytes(data)
        lib.emu_last_error.restype = ctypes.c_char_p
        rc = lib.emu_encode_fast_device(data, ctypes.c_size_t(len(data)), cfg[0], cfg[1], cfg[2], ctypes.c_size_t(cap), ctypes.byref(dst),
                                        ctypes.byref(n))
        if rc != 0:
            return None, lib.emu_last_error().decode()
        out = ctypes.string_at(dst, n.value)
        lib.emu_free(dst)
        return out, None

    encode.lib = lib
    encode.fast_device = encode_fast_device
    encode.fast = encode_fast
    encode.fast_reused = encode_fast_reused
    return encode


@pytest.fixture(scope="session")
def gpu_encoder_factory():
    if not _have_gpu():
        pytest.fail("GPU test selected but liborz_hip.so found no HIP device (no CPU fallback exists)")
    import orz_amd

    cache = {}

    def get(level):
        if level not in cache:
            cache[level] = orz_amd.StreamEncoder(device=0, level=level, mode="exact")
        return cache[level]

    yield get
    for e in cache.values():
        e.close()
