"""ctypes access to the CPU parity oracle (oracle/liborz_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by orz_amd/."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liborz_oracle.so")
CLI = os.path.join(ORACLE_DIR, "orz_oracle")

LEVELS = {0: (5, 3, 2), 1: (15, 9, 6), 2: (45, 27, 18)}  # src/main.rs:97-102


class Cfg(ctypes.Structure):
    _fields_ = [("a", ctypes.c_size_t), ("b", ctypes.c_size_t), ("c", ctypes.c_size_t)]


class Item(ctypes.Structure):
    _fields_ = [
        ("pos", ctypes.c_uint32), ("symbol", ctypes.c_uint16), ("rank", ctypes.c_uint16), ("ctx", ctypes.c_uint16),
        ("reduced_offset", ctypes.c_uint16), ("unlikely", ctypes.c_uint8), ("match_len", ctypes.c_uint8),
        ("enc_len", ctypes.c_uint8), ("after_literal", ctypes.c_uint8),
    ]


class Trace(ctypes.Structure):
    _fields_ = [("items", ctypes.POINTER(Item)), ("cap", ctypes.c_size_t), ("n", ctypes.c_size_t)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.orc_encode_mem.restype = ctypes.c_int
        _lib.orc_decode_mem.restype = ctypes.c_int
        _lib.orc_huffman_lengths.restype = ctypes.c_int
        _lib.orc_coder_selftest.restype = ctypes.c_long
        _lib.orc_hash_entry.restype = ctypes.c_uint32
    return _lib


def encode(data, level=1, cfg=None, trace_cap=0):
    """bytes -> orz stream with the oracle's restatement of orz::encode (src/lib.rs:58-92)."""
    L = lib()
    c = Cfg(*(cfg if cfg is not None else LEVELS[level]))
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    tr = None
    if trace_cap:
        buf = (Item * trace_cap)()
        tr = Trace(buf, trace_cap, 0)
    data = bytes(data)
    rc = L.orc_encode_mem(data, ctypes.c_size_t(len(data)), ctypes.byref(c), ctypes.byref(dst), ctypes.byref(n),
                          ctypes.byref(tr) if tr is not None else None)
    assert rc == 0
    out = ctypes.string_at(dst, n.value)
    L.orc_free(dst)
    if trace_cap:
        return out, [tr.items[i] for i in range(min(tr.n, trace_cap))]
    return out


def decode(stream):
    """orz stream -> bytes (src/lib.rs:94-129).  Returns (data, consumed) ; raises on InvalidData."""
    L = lib()
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    used = ctypes.c_size_t()
    stream = bytes(stream)
    rc = L.orc_decode_mem(stream, ctypes.c_size_t(len(stream)), ctypes.byref(dst), ctypes.byref(n), ctypes.byref(used))
    if rc != 0:
        raise ValueError("oracle decode: invalid data")
    out = ctypes.string_at(dst, n.value)
    L.orc_free(dst)
    return out, used.value


class PlanItem(ctypes.Structure):  # orc_plan_item
    _fields_ = [("pos", ctypes.c_uint64), ("src", ctypes.c_uint64), ("type", ctypes.c_uint8), ("len", ctypes.c_uint8)]


class PlanError(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_size_t), ("code", ctypes.c_int)]


P = (1 << 25) // 2 - 1  # SBVEC_PREMATCH_LEN
NEW = 1 << 24


def plan_from_trace(trace, n):
    """parse of a device encoder (orz_stream_get_item_trace: block, window offsets) -> plan in stream offsets"""
    plan = (PlanItem * max(1, len(trace)))()
    for i, it in enumerate(trace):
        base = int(it["block"]) * NEW
        plan[i].pos = base + int(it["pos"]) - P
        is_match = bool(int(it["after_literal"]) & 2)
        plan[i].type = 2 if is_match else (0 if int(it["symbol"]) == 388 else 1)
        plan[i].len = int(it["match_len"]) if is_match else 0
        plan[i].src = base + int(it["src"]) - P if is_match else 0
    return plan, len(trace)


def encode_plan(data, plan_n):
    """the reference's state machine + emit half applied to a given parse (orc_encode_plan_mem); raises if the
    format cannot express an item"""
    plan, nplan = plan_n
    L = lib()
    L.orc_encode_plan_mem.restype = ctypes.c_int
    dst = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    err = PlanError()
    data = bytes(data)
    rc = L.orc_encode_plan_mem(data, ctypes.c_size_t(len(data)), plan, ctypes.c_size_t(nplan), ctypes.byref(dst), ctypes.byref(n),
                               None, ctypes.byref(err))
    if rc != 0:
        raise ValueError("oracle plan encoder rejected the item at stream offset %d (code %d)" % (err.pos, err.code))
    out = ctypes.string_at(dst, n.value)
    L.orc_free(dst)
    return out


class Diag(ctypes.Structure):  # orc_diag (oracle/orz_diag.c)
    _fields_ = [("kind", ctypes.c_int32), ("cause", ctypes.c_int32), ("stream_off", ctypes.c_uint64), ("item_index", ctypes.c_uint64)] + [
        (k, ctypes.c_uint32) for k in ("block", "chunk", "item_in_chunk", "spos", "type", "symbol_rank", "symbol", "ctx", "after_literal",
                                       "unlikely", "reduced_offset", "node", "node_pos", "node_len_min", "node_len_expected", "enc_len",
                                       "match_len", "true_lcp", "src_ctx", "word0", "word1", "want0", "want1", "first_bad", "ring_count")] + [
        (k, ctypes.c_uint32 * 33) for k in ("near_pos", "near_lcp", "near_exp", "near_min")] + [
        (k, ctypes.c_uint32) for k in ("lit_rank", "word_rank", "unl_index", "tab_cnt", "tab_sum")] + [("tab_near", ctypes.c_uint32 * 9)] + [
        (k, ctypes.c_uint32) for k in ("tab_raw_index", "best_ro", "best_pos", "best_lcp", "best_exp", "best_min")]


_diag = None


def diag(stream, expect):
    """forensic decode (oracle/orz_diag.c): {} when `stream` decodes to `expect`, else the first item that does not --
    everything the decoder knew about it"""
    global _diag
    if _diag is None:
        so = os.path.join(ORACLE_DIR, "liborz_diag.so")
        if not os.path.exists(so):
            build()
        _diag = ctypes.CDLL(so)
        _diag.orc_diag_decode.restype = ctypes.c_int
    d = Diag()
    stream, expect = bytes(stream), bytes(expect)
    kind = _diag.orc_diag_decode(stream, ctypes.c_size_t(len(stream)), expect, ctypes.c_size_t(len(expect)), ctypes.byref(d))
    if kind == 0:
        return {}
    return {k: (list(getattr(d, k)) if k.startswith("near_") or k == "tab_near" else int(getattr(d, k))) for k, _ in Diag._fields_}


def assert_decodes_to(stream, expect, what="stream"):
    """the oracle's decoder must turn `stream` into `expect`; on failure the assertion names the first item no decoder
    follows (forensic decode, oracle/orz_diag.c)"""
    stream, expect = bytes(stream), bytes(expect)
    try:
        back, used = decode(stream)
        ok = back == expect and used == len(stream)
    except ValueError:
        ok = False
    if not ok:
        d = diag(stream, expect)
        raise AssertionError("%s (%d bytes) does not decode to its input (%d bytes); first wrong item: %r" % (
            what, len(stream), len(expect), {k: v for k, v in d.items() if not k.startswith("near_") and k != "tab_near"}))
