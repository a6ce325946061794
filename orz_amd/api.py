"""Host-side mirror of the reference's encoder interface, driving liborz_hip.so.

  cfg_for_level   src/main.rs:97-102
  LZEncoder       src/lz.rs:69-95   (new / encode / forward, one chunk per encode() call)
  encode          src/lib.rs:58-92  (Read -> Write stream encode)
  StreamEncoder / encode_bytes : reusable whole-buffer encoder (what bench.py times)
"""
import ctypes

from . import _native
from ._native import EncodeStats, LZCfg


class OrzError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise OrzError("%s failed (%d): %s" % (what, rc, _native.last_error()))


def cfg_for_level(level):
    """level -> LZCfg exactly as `orz encode -l` maps it (src/main.rs:97-102); other levels raise."""
    cfg = LZCfg()
    rc = _native.load().orz_lzcfg_from_level(int(level), ctypes.byref(cfg))
    if rc != 0:
        raise ValueError("invalid level")
    return cfg


def stream_bound(nbytes):
    """device bytes that always hold the stream of `nbytes` input bytes (orz_stream_bound)"""
    return int(_native.load().orz_stream_bound(int(nbytes)))


class OrzBuffer:
    """A stream the library returned (malloc'ed by orz_stream_encode), held without copying; freed with orz_free."""

    def __init__(self, lib, ptr, n):
        self._lib, self._ptr, self._n = lib, ptr, int(n)

    def __len__(self):
        return self._n

    def __bytes__(self):
        return ctypes.string_at(self._ptr, self._n)

    def view(self):
        """memoryview over the buffer (valid while this object lives)"""
        return memoryview((ctypes.c_uint8 * self._n).from_address(ctypes.addressof(self._ptr.contents))).cast("B") if self._n else memoryview(b"")

    def close(self):
        if self._ptr is not None:
            self._lib.orz_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamEncoder:
    """One orz stream encoder bound to one GPU; reusable across inputs."""

    def __init__(self, device=0, level=1, cfg=None, mode=None, tile_bytes=0, rounds=0):
        """mode: "fast" (GPU-native parse: reference-decodable, size within +-0.5 %), "exact" (the reference's
        parse item for item: byte-identical stream) or None = the library default (fast unless ORZ_MODE=exact)."""
        self._lib = _native.load()
        self.cfg = cfg if cfg is not None else cfg_for_level(level)
        self._h = self._lib.orz_stream_new(int(device), ctypes.byref(self.cfg))
        if not self._h:
            raise OrzError("orz_stream_new failed: " + _native.last_error())
        if mode is not None or tile_bytes or rounds:
            if mode is None:
                mode = "fast" if self.config()["mode"] == 1 else "exact"
            if mode not in ("fast", "exact"):
                raise ValueError("mode must be 'fast' or 'exact'")
            _check(self._lib.orz_stream_set_mode(self._h, 1 if mode == "fast" else 0, int(tile_bytes), int(rounds)),
                   "orz_stream_set_mode")

    def set_mode(self, mode, tile_bytes=0, rounds=0):
        """switch the parse mode / fast-mode schedule of this encoder (rebuilds its device state)"""
        if mode not in ("fast", "exact"):
            raise ValueError("mode must be 'fast' or 'exact'")
        _check(self._lib.orz_stream_set_mode(self._h, 1 if mode == "fast" else 0, int(tile_bytes), int(rounds)), "orz_stream_set_mode")

    def config(self):
        """What the encoder runs with (orz_stream_get_config)."""
        c = _native.StreamConfig()
        _check(self._lib.orz_stream_get_config(self._h, ctypes.byref(c)), "orz_stream_get_config")
        return c.as_dict()

    def set_profile(self, on=True):
        """bracket the kernels inside the round loop too (no hipGraph replay then): for the roofline leg only"""
        _check(self._lib.orz_stream_set_profile(self._h, 1 if on else 0), "orz_stream_set_profile")

    def kernel_times(self):
        """[(ms, launches)] x 4 of the last encode(stats=True): parse kernel, symbol ranking, candidate tables, path maps"""
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_uint64 * 4)()
        _check(self._lib.orz_stream_get_kernel_times(self._h, ms, n), "orz_stream_get_kernel_times")
        return [(ms[i], n[i]) for i in range(4)]

    def kernel_table(self):
        """[(name, ms, launches)] of EVERY kernel of the last encode(stats=True) made in profile mode, largest first"""
        n = self._lib.orz_stream_get_kernel_table(self._h, None, 0)
        if n <= 0:
            return []
        rows = (_native.KernelRow * n)()
        self._lib.orz_stream_get_kernel_table(self._h, ctypes.cast(rows, ctypes.c_void_p), n)
        return [(r.name.decode(), r.ms, int(r.launches)) for r in rows]

    def set_tuning(self, seg_bytes=0, window_segs=0):
        _check(self._lib.orz_stream_set_tuning(self._h, seg_bytes, window_segs), "orz_stream_set_tuning")

    def _encode(self, ptr, n, on_device, want_stats, raw=False):
        dst = ctypes.POINTER(ctypes.c_uint8)()
        dlen = ctypes.c_size_t()
        st = EncodeStats()
        rc = self._lib.orz_stream_encode(
            self._h, ptr, n, 1 if on_device else 0, ctypes.byref(dst), ctypes.byref(dlen),
            ctypes.byref(st) if want_stats else None,
        )
        _check(rc, "orz_stream_encode")
        if raw:  # the library's buffer itself, no copy (released with the object)
            return OrzBuffer(self._lib, dst, dlen.value), (st.as_dict() if want_stats else None)
        try:
            out = ctypes.string_at(dst, dlen.value)
        finally:
            self._lib.orz_free(dst)
        return out, (st.as_dict() if want_stats else None)

    def encode(self, data, stats=False):
        """bytes -> orz stream (same bytes `orz encode` writes)."""
        data = bytes(data)
        buf = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
        out, st = self._encode(ctypes.cast(buf, ctypes.c_void_p), len(data), False, stats)
        return (out, st) if stats else out

    def encode_device(self, dev_ptr, nbytes, stats=False, raw=False):
        """Encode `nbytes` already resident in this GPU's HBM at address `dev_ptr`.  raw=True returns the library's own
        host buffer (an OrzBuffer: len(), bytes(), .view() -> memoryview; it does not implement the buffer protocol itself)
        instead of a bytes copy of it."""
        out, st = self._encode(ctypes.c_void_p(int(dev_ptr)), int(nbytes), True, stats, raw)
        return (out, st) if stats else out

    def encode_to_device(self, src_ptr, nbytes, dst_ptr, dst_cap, src_on_device=True, stats=False):
        """Encode into DEVICE memory the caller owns (orz_stream_encode_to_device): the finished stream -- framed on the
        device -- is left at `dst_ptr` (`dst_cap` bytes on this encoder's GPU; `stream_bound(nbytes)` always suffices).
        Returns its length (and the stats dict).  What the multi-GPU gather sends from where it lies."""
        dlen = ctypes.c_size_t()
        st = EncodeStats()
        rc = self._lib.orz_stream_encode_to_device(self._h, ctypes.c_void_p(int(src_ptr)), int(nbytes), 1 if src_on_device else 0,
                                                   ctypes.c_void_p(int(dst_ptr)), int(dst_cap), ctypes.byref(dlen),
                                                   ctypes.byref(st) if stats else None)
        _check(rc, "orz_stream_encode_to_device")
        return (dlen.value, st.as_dict()) if stats else dlen.value

    def set_item_trace(self, on=True):
        _check(self._lib.orz_stream_set_item_trace(self._h, 1 if on else 0), "orz_stream_set_item_trace")

    def item_trace(self):
        """numpy structured array of the items of the last encode() (needs set_item_trace(True))."""
        import numpy as np

        n = self._lib.orz_stream_get_item_trace(self._h, None, 0)
        buf = (_native.Item * max(n, 1))()
        self._lib.orz_stream_get_item_trace(self._h, buf, n)
        dt = np.dtype([("block", "<u4"), ("pos", "<u4"), ("symbol", "<u2"), ("rank", "<u2"), ("ctx", "<u2"),
                       ("robits", "<u2"), ("unlikely", "u1"), ("enc_len", "u1"), ("after_literal", "u1"), ("match_len", "u1"),
                       ("src", "<u4")])
        return np.frombuffer(buf, dtype=dt, count=n).copy()

    def close(self):
        if self._h:
            self._lib.orz_stream_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_bytes(data, level=1, device=0):
    enc = StreamEncoder(device=device, level=level)
    try:
        return enc.encode(data)
    finally:
        enc.close()


def encode(src, dst, cfg, progress=None, device=0):
    """orz::encode (src/lib.rs:58-92): read everything from file-like `src`, write the stream to `dst`.

    `progress(is_finish, in_bytes, out_bytes)` mirrors ProgressLogger (src/progress.rs:9-13).
    Returns (bytes_read, bytes_written) like the reference's CountRead/CountWrite totals."""
    lib = _native.load()
    counts = [0, 0]

    def _rd(_ctx, buf, cap):
        try:
            chunk = src.read(cap)
        except Exception:
            return -1
        n = len(chunk)
        if n:
            ctypes.memmove(buf, chunk, n)
            counts[0] += n
        return n

    def _wr(_ctx, buf, n):
        try:
            dst.write(ctypes.string_at(buf, n))
        except Exception:
            return -1
        counts[1] += n
        return 0

    def _pg(_ctx, fin, a, b):
        if progress:
            progress(bool(fin), a, b)

    rd, wr, pg = _native.READ_FN(_rd), _native.WRITE_FN(_wr), _native.PROGRESS_FN(_pg)
    _check(lib.orz_encode(rd, None, wr, None, ctypes.byref(cfg), pg, None, int(device)), "orz_encode")
    return tuple(counts)


class MemberEncoder:
    """`jobs` stream encoders on one GPU; encode() cuts the input into members of `member_bytes`, encodes
    them concurrently (one host thread per encoder inside the library) and returns the members' streams
    concatenated in order -- each a complete orz stream the reference decoder reads."""

    def __init__(self, device=0, level=1, jobs=4, devices=None):
        """devices: list of HIP ordinals for a multi-GPU job (`jobs` encoders on each); default = [device]"""
        self._lib = _native.load()
        self.cfg = cfg_for_level(level)
        devs = [int(device)] if devices is None else [int(d) for d in devices]
        arr = (ctypes.c_int * len(devs))(*devs)
        self._h = self._lib.orz_members_new_multi(arr, len(devs), ctypes.byref(self.cfg), int(jobs))
        if not self._h:
            raise OrzError("orz_members_new_multi failed: " + _native.last_error())

    def _run(self, ptr, n, on_device, member_bytes):
        dst = ctypes.POINTER(ctypes.c_uint8)()
        dlen, nm = ctypes.c_size_t(), ctypes.c_size_t()
        rc = self._lib.orz_members_encode(self._h, ptr, n, 1 if on_device else 0, int(member_bytes), ctypes.byref(dst),
                                          ctypes.byref(dlen), ctypes.byref(nm))
        _check(rc, "orz_members_encode")
        try:
            return ctypes.string_at(dst, dlen.value), nm.value
        finally:
            self._lib.orz_free(dst)

    def encode(self, data, member_bytes=1 << 26):
        """member_bytes: 64 MiB by default -- a member starts with empty rings and a flat symbol order, and that cold start
        costs about 1.5 % of the size at 16 MiB members, a quarter of that at 64 MiB.  Members larger than one
        block (16 MiB) decode with the reference decoder, with `decode_members` (host) and with the device decoder
        (`decode_members_device`: its window slides as a counter, round 4)."""
        data = bytes(data)  # (no copy when it is `bytes` already; the library reads the object's own buffer)
        ptr = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p) if data else ctypes.cast(ctypes.create_string_buffer(1), ctypes.c_void_p)
        return self._run(ptr, len(data), False, member_bytes)

    def encode_device(self, dev_ptr, nbytes, member_bytes=1 << 26):
        return self._run(ctypes.c_void_p(int(dev_ptr)), int(nbytes), True, member_bytes)

    def encode_to_device(self, src_ptr, nbytes, dst_ptr, dst_cap, member_bytes=1 << 26, src_on_device=True):
        """The members' streams left in DEVICE memory the caller owns (orz_members_encode_to_device; one GPU): returns
        [(offset, length)] per member, in member order, into the buffer at `dst_ptr`."""
        nm = 1 if nbytes == 0 else (int(nbytes) + int(member_bytes) - 1) // int(member_bytes)
        offs, lens = (ctypes.c_size_t * nm)(), (ctypes.c_size_t * nm)()
        got = ctypes.c_size_t()
        rc = self._lib.orz_members_encode_to_device(self._h, ctypes.c_void_p(int(src_ptr)), int(nbytes), 1 if src_on_device else 0,
                                                    int(member_bytes), ctypes.c_void_p(int(dst_ptr)), int(dst_cap), offs, lens,
                                                    ctypes.byref(got))
        _check(rc, "orz_members_encode_to_device")
        return [(offs[k], lens[k]) for k in range(got.value)]

    def close(self):
        if self._h:
            self._lib.orz_members_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_members(container):
    """decode every stream of a concatenation of orz streams -> (bytes, n_members)"""
    lib = _native.load()
    container = bytes(container)
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n, nm = ctypes.c_size_t(), ctypes.c_size_t()
    buf = ctypes.create_string_buffer(container, len(container)) if container else ctypes.create_string_buffer(1)
    rc = lib.orz_decode_members_mem(ctypes.cast(buf, ctypes.c_void_p), len(container), ctypes.byref(dst), ctypes.byref(n),
                                    ctypes.byref(nm))
    _check(rc, "orz_decode_members_mem")
    try:
        return ctypes.string_at(dst, n.value), nm.value
    finally:
        lib.orz_free(dst)


def decode_members_device(container, device=0, stats=False):
    """decode every member of a concatenation of orz streams ON THE GPU (one member per wavefront; members of any number
    of blocks below 4 GiB -- the window slides as a counter, round 4) -> (bytes, n_members[, stats dict]).  Same bytes as
    `decode_members`."""
    lib = _native.load()
    container = bytes(container)
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n, nm = ctypes.c_size_t(), ctypes.c_size_t()
    st = _native.DecodeStats()
    buf = ctypes.create_string_buffer(container, len(container)) if container else ctypes.create_string_buffer(1)
    rc = lib.orz_decode_members_device(device, ctypes.cast(buf, ctypes.c_void_p), len(container), ctypes.byref(dst),
                                       ctypes.byref(n), ctypes.byref(nm), ctypes.byref(st))
    _check(rc, "orz_decode_members_device")
    try:
        out = ctypes.string_at(dst, n.value)
    finally:
        lib.orz_free(dst)
    return (out, nm.value, st.as_dict()) if stats else (out, nm.value)


def huffman_tables(weights, device=0):
    """Huffman code lengths and canonical codes of every table of `weights` ON THE GPU -- an array of shape
    [nchunks, orz_huffman_stride()] of symbol weights below 2^23 in the encoder's layout (389 + 389 + 240 symbols a chunk;
    HuffmanTable::new_from_sym_weights + HuffmanEncoding::from_huffman_table, src/huffman.rs:27-141)
    -> (lens uint8 array, codes uint16 array, microseconds of the kernel launch)."""
    import numpy as np

    lib = _native.load()
    w0 = np.asarray(weights)
    if w0.size and (not np.issubdtype(w0.dtype, np.integer) or (w0 < 0).any() or (w0 >= (1 << 23)).any()):
        raise ValueError("weights must be integers in [0, 2^23)")  # (before the cast: a negative or huge weight must not wrap into range)
    w = np.ascontiguousarray(w0, dtype=np.uint32)
    stride = lib.orz_huffman_stride()
    if w.ndim != 2 or w.shape[1] != stride:
        raise ValueError("weights must have shape [nchunks, %d]" % stride)
    lens = np.zeros(w.shape, dtype=np.uint8)
    codes = np.zeros(w.shape, dtype=np.uint16)
    us = ctypes.c_double()
    rc = lib.orz_huffman_tables(device, w.ctypes.data, w.shape[0], lens.ctypes.data, codes.ctypes.data, ctypes.byref(us))
    _check(rc, "orz_huffman_tables")
    return lens, codes, us.value


def decode_bytes(stream):
    """orz stream -> (bytes, consumed).  Host decoder of the library (orz::decode, src/lib.rs:94-129);
    stops after the first stream's EOF chunk like the reference."""
    lib = _native.load()
    stream = bytes(stream)
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n, used = ctypes.c_size_t(), ctypes.c_size_t()
    buf = ctypes.create_string_buffer(stream, len(stream)) if stream else ctypes.create_string_buffer(1)
    rc = lib.orz_decode_mem(ctypes.cast(buf, ctypes.c_void_p), len(stream), ctypes.byref(dst), ctypes.byref(n), ctypes.byref(used))
    _check(rc, "orz_decode_mem")
    try:
        return ctypes.string_at(dst, n.value), used.value
    finally:
        lib.orz_free(dst)


def decode(src, dst, progress=None):
    """orz::decode (src/lib.rs:94-129) between file-like objects; returns (bytes_read, bytes_written)."""
    lib = _native.load()
    counts = [0, 0]

    def _rd(_ctx, buf, cap):
        try:
            chunk = src.read(cap)
        except Exception:
            return -1
        if chunk:
            ctypes.memmove(buf, chunk, len(chunk))
            counts[0] += len(chunk)
        return len(chunk)

    def _wr(_ctx, buf, n):
        try:
            dst.write(ctypes.string_at(buf, n))
        except Exception:
            return -1
        counts[1] += n
        return 0

    def _pg(_ctx, fin, a, b):
        if progress:
            progress(bool(fin), a, b)

    rd, wr, pg = _native.READ_FN(_rd), _native.WRITE_FN(_wr), _native.PROGRESS_FN(_pg)
    _check(lib.orz_decode(rd, None, wr, None, pg, None), "orz_decode")
    return tuple(counts)


class LZEncoder:
    """Object-level mirror of the reference's LZEncoder (src/lz.rs:69-95).

    `encode(cfg, window, sbuf_len, spos)` takes the caller's window allocation INCLUDING the two
    480-byte sentinel pads (i.e. `window[480]` is sbuf[0], as `orz::encode` lays it out,
    src/lib.rs:67-69) and returns (spos_out, chunk_bytes)."""

    def __init__(self, device=0):
        self._lib = _native.load()
        self._h = self._lib.orz_lz_encoder_new(int(device))
        if not self._h:
            raise OrzError("orz_lz_encoder_new failed: " + _native.last_error())
        self._tbuf = ctypes.create_string_buffer(3 * 16777215)

    def encode(self, cfg, window, sbuf_len, spos):
        base = ctypes.addressof(window) if not isinstance(window, int) else window
        so, tl = ctypes.c_size_t(), ctypes.c_size_t()
        rc = self._lib.orz_lz_encoder_encode(
            self._h, ctypes.byref(cfg), ctypes.c_void_p(base + 480), sbuf_len, self._tbuf, len(self._tbuf), spos,
            ctypes.byref(so), ctypes.byref(tl),
        )
        _check(rc, "orz_lz_encoder_encode")
        return so.value, self._tbuf.raw[: tl.value]

    def forward(self, forward_len):
        _check(self._lib.orz_lz_encoder_forward(self._h, forward_len), "orz_lz_encoder_forward")

    def close(self):
        if self._h:
            self._lib.orz_lz_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
