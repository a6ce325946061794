"""orz_amd -- MI355X-native ROLZ encoder with the bitstream and call surface of richox/orz.

Python host mirror of the reference's public interface (src/lib.rs:22-24,58-63; src/lz.rs:32-47):
`LZCfg`, `encode`, plus the object-level `LZEncoder`.  All compute happens in liborz_hip.so (HIP
kernels for gfx950 behind the C ABI of include/orz_hip.h).
"""
from ._native import LIB_PATH, EncodeStats, LZCfg  # noqa: F401
from .api import (LZEncoder, OrzError, MemberEncoder, StreamEncoder, cfg_for_level, decode, decode_bytes,  # noqa: F401
                  decode_members, decode_members_device, encode, encode_bytes, huffman_tables, stream_bound)

LZ_BLOCK_SIZE = (1 << 25) - 1  # src/lib.rs:31
SBVEC_SENTINEL_LEN = 480  # src/lib.rs:54
SBVEC_PREMATCH_LEN = LZ_BLOCK_SIZE // 2  # src/lib.rs:55
