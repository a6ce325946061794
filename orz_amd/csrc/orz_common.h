// orz_common.h -- constants and byte-level primitives shared by every kernel of the MI355X
// ROLZ encoder.  All functions are ORZ_HD: the same source is compiled by hipcc for gfx950
// (product) and by g++ for the host emulation backend used by the CPU-only tests.
//
// Reference interfaces restated here (no code shared with the reference, which is Rust):
//   constants      /root/reference/src/lib.rs:31-34,54-55, src/lz.rs:24-29, src/matcher.rs:18
//   hash1 / hash2  src/lz.rs:482-492
//   hash_dword     src/matcher.rs:256-263
//   LCP            src/mem.rs:41-51
//   ROID tables    src/lz.rs:494-534
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORZ_HD __host__ __device__ __forceinline__
#define ORZ_D __device__ __forceinline__
#else
#define ORZ_HD inline
#define ORZ_D inline
#endif

namespace orz {

constexpr uint32_t kBlock = (1u << 25) - 1;       // LZ_BLOCK_SIZE, src/lib.rs:31
constexpr uint32_t kPre = kBlock / 2;             // SBVEC_PREMATCH_LEN = 16,777,215, src/lib.rs:55
constexpr uint32_t kNewMax = kBlock - kPre;       // 16,777,216 new bytes per block
constexpr uint32_t kSent = 480;                   // SBVEC_SENTINEL_LEN, src/lib.rs:54
constexpr uint32_t kChunkItems = 1u << 20;        // LZ_CHUNK_SIZE, src/lib.rs:32
constexpr uint32_t kMaxLen = 240, kMinLen = 4;    // src/lib.rs:33-34
constexpr uint32_t kRing = 4094;                  // LZ_MF_BUCKET_ITEM_SIZE, src/lz.rs:24
constexpr uint32_t kHash = 4627;                  // LZ_MF_BUCKET_ITEM_HASH_SIZE, src/matcher.rs:18
constexpr uint32_t kSyms = 389;                   // SYMRANK_NUM_SYMBOLS, src/lz.rs:25
constexpr uint32_t kWordSym = 388;                // WORD_SYMBOL, src/lz.rs:29
constexpr uint32_t kLenSyms = 240;                // huff_weights2 size, src/lz.rs:273
constexpr uint32_t kWLen = 1u << 25;              // per-position array length (window offsets)
constexpr int kPosBits = 25;
constexpr uint64_t kPosMask = (1ull << kPosBits) - 1;

// item types stored in TY[] (low 2 bits) ; bit 2 = after_literal at the item start
enum : uint8_t { kTyWord = 0, kTyLit = 1, kTyMatch = 2 };

ORZ_HD uint32_t ld32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
ORZ_HD int is_alnum(uint8_t c) {
    return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
}
// ctx8 of the item starting at x is hash1(x-1): src/lz.rs:135,140,482-486
ORZ_HD uint32_t hash1(const uint8_t* b, uint32_t pos) {
    return (uint32_t)(b[pos] & 0x7f) | ((uint32_t)is_alnum(b[(int64_t)pos - 1]) << 7);
}
ORZ_HD uint32_t hash2(const uint8_t* b, uint32_t pos) {  // src/lz.rs:489-492
    return (uint32_t)(b[pos] & 0x7f) | (hash1(b, pos - 1) << 7);
}
ORZ_HD uint32_t hash_entry(const uint8_t* p) {  // hash_dword % 4627, src/matcher.rs:117,256-263
    uint32_t h = ((uint32_t)p[0] * 131313131u ^ 797u) + ((uint32_t)p[1] * 1313131u ^ 79797u) +
                 ((uint32_t)p[2] * 13131u ^ 7979797u) + ((uint32_t)p[3] * 131u ^ 797979797u);
    return h % kHash;
}
// bucket key of position x: (ctx8, hash entry) -> 21 bits
ORZ_HD uint32_t bucket_key(const uint8_t* b, uint32_t x) { return hash1(b, x - 1) * kHash + hash_entry(b + x); }

// Longest common prefix of b[p1..] and b[p2..], capped at 240 (src/mem.rs:41-51 gives the same
// value: it compares 16-byte lanes and returns the first differing byte index, cap 240).
ORZ_HD uint32_t lcp240(const uint8_t* b, uint32_t p1, uint32_t p2) {
    uint32_t l = 0;
    while (l < kMaxLen) {
        uint32_t a = ld32(b + p1 + l), c = ld32(b + p2 + l);
        if (a != c) {
            uint32_t x = a ^ c;
            uint32_t k = (x & 0xff) ? 0 : (x & 0xff00) ? 1 : (x & 0xff0000) ? 2 : 3;
            return l + k;
        }
        l += 4;
    }
    return kMaxLen;
}

// reduced offset -> (roid, extra bit count, extra bits), src/lz.rs:494-514.  roid i covers
// 2^(i/2) offsets; bases 0,1,2,4,6,10,14,22,30,46,62,94,126,190,254,382,510,766,1022,1534,2046,3070.
ORZ_HD void roid_encode(uint32_t ro, uint32_t* roid, uint32_t* bitlen, uint32_t* bits) {
    uint32_t base = 0, id = 0;
    for (;;) {
        uint32_t n = 1u << (id >> 1);
        if (ro < base + n) {
            *roid = id;
            *bitlen = id >> 1;
            *bits = ro - base;
            return;
        }
        base += n;
        id++;
    }
}
ORZ_HD uint32_t roid_bitlen(uint32_t ro) {
    uint32_t a, b, c;
    roid_encode(ro, &a, &b, &c);
    return b;
}

// level -> LZCfg, src/main.rs:97-102
struct Cfg {
    int depth, lazy1, lazy2;
};

}  // namespace orz
