"""ctypes binding of liborz_hip.so (C ABI in include/orz_hip.h).

The library is the product; there is deliberately no Python or CPU fallback: if the shared object
is missing or no HIP device is usable, constructing an encoder raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORZ_LIB_PATH") or os.path.join(_HERE, "lib", "liborz_hip.so")  # (ORZ_LIB_PATH: dev builds side by side)


class LZCfg(ctypes.Structure):
    """#[repr(C)] LZCfg of the reference (src/lz.rs:32-37): three usize."""

    _fields_ = [
        ("match_depth", ctypes.c_size_t),
        ("lazy_match_depth1", ctypes.c_size_t),
        ("lazy_match_depth2", ctypes.c_size_t),
    ]


class EncodeStats(ctypes.Structure):
    _fields_ = [
        ("blocks", ctypes.c_uint64),
        ("sweeps", ctypes.c_uint64),
        ("seg_evals", ctypes.c_uint64),
        ("items", ctypes.c_uint64),
        ("chunks", ctypes.c_uint64),
        ("in_bytes", ctypes.c_uint64),
        ("out_bytes", ctypes.c_uint64),
        ("t_prep_s", ctypes.c_double),
        ("t_parse_s", ctypes.c_double),
        ("t_post_s", ctypes.c_double),
        ("parse_kernel_ms", ctypes.c_double),
        ("parse_launches", ctypes.c_uint64),
        ("total_ms", ctypes.c_double),
        ("host_syncs", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KernelRow(ctypes.Structure):  # orz_kernel_row
    _fields_ = [("name", ctypes.c_char * 64), ("ms", ctypes.c_double), ("launches", ctypes.c_uint64)]


class StreamConfig(ctypes.Structure):  # orz_stream_config
    _fields_ = [
        ("mode", ctypes.c_int), ("segment_bytes", ctypes.c_uint), ("window_segments", ctypes.c_uint),
        ("fast_tile_bytes", ctypes.c_uint), ("fast_rounds", ctypes.c_uint), ("fast_row_entries", ctypes.c_uint),
        ("unit_bytes", ctypes.c_uint),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Item(ctypes.Structure):
    _fields_ = [
        ("block", ctypes.c_uint32), ("pos", ctypes.c_uint32), ("symbol", ctypes.c_uint16), ("rank", ctypes.c_uint16),
        ("ctx", ctypes.c_uint16), ("robits", ctypes.c_uint16), ("unlikely", ctypes.c_uint8), ("enc_len", ctypes.c_uint8),
        ("after_literal", ctypes.c_uint8), ("match_len", ctypes.c_uint8), ("src", ctypes.c_uint32),
    ]


READ_FN = ctypes.CFUNCTYPE(ctypes.c_ssize_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)
WRITE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)
PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t)

# every symbol include/orz_hip.h declares: (name, restype, argtypes)
class DecodeStats(ctypes.Structure):  # orz_decode_stats
    _fields_ = [
        ("members", ctypes.c_uint64),
        ("in_bytes", ctypes.c_uint64),
        ("out_bytes", ctypes.c_uint64),
        ("launches", ctypes.c_uint64),
        ("kernel_ms", ctypes.c_double),
        ("total_s", ctypes.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


SYMBOLS = [
    ("orz_lzcfg_from_level", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(LZCfg)]),
    ("orz_lz_encoder_new", ctypes.c_void_p, [ctypes.c_int]),
    ("orz_lz_encoder_free", None, [ctypes.c_void_p]),
    (
        "orz_lz_encoder_encode",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.POINTER(LZCfg), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
         ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)],
    ),
    ("orz_lz_encoder_forward", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    (
        "orz_encode",
        ctypes.c_int,
        [READ_FN, ctypes.c_void_p, WRITE_FN, ctypes.c_void_p, ctypes.POINTER(LZCfg), PROGRESS_FN, ctypes.c_void_p,
         ctypes.c_int],
    ),
    ("orz_lz_decoder_new", ctypes.c_void_p, []),
    ("orz_lz_decoder_free", None, [ctypes.c_void_p]),
    (
        "orz_lz_decoder_decode",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)],
    ),
    ("orz_lz_decoder_forward", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    ("orz_decode", ctypes.c_int, [READ_FN, ctypes.c_void_p, WRITE_FN, ctypes.c_void_p, PROGRESS_FN, ctypes.c_void_p]),
    (
        "orz_decode_mem",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t),
         ctypes.POINTER(ctypes.c_size_t)],
    ),
    ("orz_stream_new", ctypes.c_void_p, [ctypes.c_int, ctypes.POINTER(LZCfg)]),
    ("orz_stream_free", None, [ctypes.c_void_p]),
    ("orz_stream_set_tuning", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]),
    ("orz_stream_set_mode", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]),
    ("orz_stream_get_config", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StreamConfig)]),
    ("orz_stream_set_profile", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("orz_stream_get_kernel_times", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]),
    ("orz_stream_get_kernel_table", ctypes.c_long, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    (
        "orz_stream_encode",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(EncodeStats)],
    ),
    ("orz_free", None, [ctypes.c_void_p]),
    ("orz_stream_bound", ctypes.c_size_t, [ctypes.c_size_t]),
    (
        "orz_stream_encode_to_device",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(EncodeStats)],
    ),
    (
        "orz_members_encode_to_device",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)],
    ),
    ("orz_members_new_multi", ctypes.c_void_p, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(LZCfg), ctypes.c_int]),
    ("orz_members_new", ctypes.c_void_p, [ctypes.c_int, ctypes.POINTER(LZCfg), ctypes.c_int]),
    ("orz_members_free", None, [ctypes.c_void_p]),
    (
        "orz_members_encode",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t,
         ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)],
    ),
    (
        "orz_decode_members_mem",
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t),
         ctypes.POINTER(ctypes.c_size_t)],
    ),
    (
        "orz_decode_members_device",
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(DecodeStats)],
    ),
    ("orz_stream_set_item_trace", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("orz_stream_get_item_trace", ctypes.c_long, [ctypes.c_void_p, ctypes.POINTER(Item), ctypes.c_size_t]),
    ("orz_huffman_stride", ctypes.c_size_t, []),
    (
        "orz_huffman_tables",
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)],
    ),
    ("orz_device_count", ctypes.c_int, []),
    ("orz_last_error", ctypes.c_char_p, []),
    ("orz_version", ctypes.c_char_p, []),
]

_lib = None


def load():
    """Load liborz_hip.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liborz_hip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH
        )
    # (the library sets the same default when it is loaded; a host that initialised HIP earlier -- torch -- has to export it)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    return load().orz_last_error().decode("utf-8", "replace")
