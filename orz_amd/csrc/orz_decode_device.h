// orz_decode_device.h -- device-side orz decoder, one member (complete orz stream) per wavefront.
//
// SURVEY.md 8(f) row 3.  Decoding a stream is one serial chain (every symbol's context depends on the
// bytes decoded before it), so a member is decoded by ONE lane; the parallelism is across members, which a
// members container has plenty of.  The kernel follows the reference's LZDecoder::decode / orz::decode
// (/root/reference/src/lz.rs:366-478, src/lib.rs:94-129, src/coder.rs:91-217, src/huffman.rs:118-167,
// src/symrank.rs:49-97, src/matcher.rs:62-80) the way the library's host decoder (orz_host_decode.h) does,
// restated for a GPU lane: no exceptions (status codes), all model state in one zero-initialised blob of
// HBM per member, the last eight decoded bytes in a register so the three context hashes need no loads,
// canonical Huffman tables as 2^max_len-entry lookup tables in the blob.
//
// Round 6: the small hot state moved to LDS -- the words[] table (64 KB), a 12-bit first-level lookup table per Huffman table
// (3 x 8 KB; codes longer than 12 bits fall through to the full table in the blob) and the 256 ring heads: 89 KB a wavefront, so a
// CU runs one member -- and the chain of dependent memory round trips per item got shorter: the rank table's candidates
// value[r], value[r + 1] are asked for together with index[excluded symbol] (the rank i is one of the two), a match whose source
// does not overlap it is copied eight bytes a load, the bit reader refills with one 4-byte load.  Still ONE lane per member: what
// bounds it now is the rank tables (800 KB) and the rings (6.3 MB) of a member, which live in L2 / Infinity Cache -- two to four
// dependent round trips of 200..900 cycles per item.
//
// A member decodes straight into its place in the output buffer (positions before the member's first byte read as
// zero, as the reference's window does), so the window never has to be MOVED: ring nodes hold offsets into the member,
// and the reference's slide (src/lib.rs:119-124; Bucket::forward, src/matcher.rs:82-87) is a counter -- a node whose
// position has left the window (window offset <= 0 after the slides so far) is dead, as `pos = 0` is in the reference.
// Members of any number of blocks below 4 GiB decode here (round 4; before, one block: the default 64 MiB members of
// `orz_members_encode` were refused).  One difference to the reference on MALFORMED streams only: a match that names a
// dead node is rejected as invalid data here (the reference copies from window offset 0).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "orz_common.h"
#include "orz_parse.h"  // (ldu32 / ldu64 / stu64: unaligned loads and stores)

namespace orz {

enum : uint32_t {
    kDecOk = 0,
    kDecBadData = 1,     // InvalidData of the reference (src/lz.rs:413-415, src/lib.rs:111-113)
    kDecTooLarge = 2,    // (unused since round 4: members of several blocks decode here)
    kDecSizeMismatch = 3,// decoded size differs from what the chunk headers announced
    kDecDeepTable = 4    // a 16-bit Huffman table: the reference accepts it, no orz encoder writes it (src/huffman.rs:99-108
                         // caps at 15) and the lookup tables here hold 2^15 entries: use the host decoder
};

struct DecodeLayout {  // byte offsets into one member's state blob
    static constexpr size_t kRingPos = 0;
    static constexpr size_t kRingMin = kRingPos + (size_t)256 * kRing * 4;
    static constexpr size_t kRingExp = kRingMin + (size_t)256 * kRing;
    static constexpr size_t kHead = (kRingExp + (size_t)256 * kRing + 15) / 16 * 16;
    static constexpr size_t kRankVal = kHead + 256 * 4;
    static constexpr size_t kRankIdx = (kRankVal + (size_t)512 * kSyms * 2 + 15) / 16 * 16;
    static constexpr size_t kRankCnt = (kRankIdx + (size_t)512 * kSyms * 2 + 15) / 16 * 16;
    static constexpr size_t kRankSum = kRankCnt + 512 * 4;
    static constexpr size_t kWords = kRankSum + 512 * 4;
    static constexpr size_t kLut = kWords + 65536;            // 3 tables x 32768 x u16: (symbol << 4) | length
    static constexpr size_t kLens = kLut + (size_t)3 * 32768 * 2;  // scratch: code lengths being read
    static constexpr size_t kOrder = kLens + 512;              // scratch: census order (389 x u16)
    static constexpr size_t kBytes = (kOrder + 1024 + 255) / 256 * 256;
};

struct DecodeLds {  // byte offsets into a member's LDS (DecodeMember::lds_bytes())
    static constexpr uint32_t kPrimBits = 12;
    static constexpr size_t kWords = 0;                                  // [32768][2] words[]
    static constexpr size_t kPrim = kWords + 65536;                      // [3][4096] u16 first-level lookup: (symbol << 4) | length, kLong = look in the full table
    static constexpr size_t kHead = kPrim + (size_t)3 * 4096 * 2;        // [256] u32 ring heads
    static constexpr size_t kRecip = kHead + 256 * 4;                    // [64] u32 reciprocals of the steady-state counts 327 + k
    static constexpr size_t kRankCnt = kRecip + 64 * 4;                  // [512] u32 encoded_cnt of each symbol-ranking context
    static constexpr size_t kRankSum = kRankCnt + 512 * 4;               // [512] u32 encoded_idx_sum
    static constexpr size_t kBytes = kRankSum + 512 * 4;
};
constexpr uint16_t kDecLong = 0xffff;  // (no real entry: a symbol is below 512 and a length below 16)

struct DecodeArgs {
    const uint8_t* src;        // the container
    const uint64_t* m_begin;   // [members] offset of the member's first chunk-length prefix
    const uint64_t* m_end;     // [members] offset one past its EOF byte
    const uint64_t* out_off;   // [members] where the member's bytes go in `out`
    const uint32_t* out_len;   // [members] decoded size announced by its chunk headers
    uint8_t* out;
    uint8_t* state;            // [slots] DecodeLayout::kBytes each, zeroed before the launch
    uint32_t* status;          // [members] kDec*
    uint32_t first, count;     // this launch decodes members first .. first + count - 1, one per block
};

// four bytes at any address, little-endian (the bit reader runs on the host too: index_members)
ORZ_HD uint32_t dec_ld32(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ldu32(p);
#else
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
#endif
}

struct DecodeMember {
    DecodeArgs a;

    struct Bits {  // MSB-first bit reader over big-endian 32-bit words (src/coder.rs:159-217)
        const uint8_t* p;
        uint32_t n, at;
        uint64_t acc;
        uint32_t have;
        ORZ_HD void fill() {
            if (have >= 32) return;
            uint32_t w = 0;  // reading past the chunk yields zeros: the encoder padded to 32 bits
            if (at + 4 <= n) { const uint32_t le = dec_ld32(p + at); w = (le << 24) | ((le & 0xff00) << 8) | ((le >> 8) & 0xff00) | (le >> 24); }
            else for (uint32_t i = 0; i < 4; i++) w = (w << 8) | (at + i < n ? p[at + i] : 0);
            at += 4;
            acc = (acc << 32) | w;
            have += 32;
        }
        ORZ_HD uint32_t bits(uint32_t k) {  // k <= 32
            if (k == 0) return 0;
            fill();
            const uint32_t v = (uint32_t)(acc >> (have - k)) & (k == 32 ? 0xffffffffu : ((1u << k) - 1));
            have -= k;
            return v;
        }
        ORZ_HD uint32_t peek(uint32_t k) {
            if (k == 0) return 0;
            fill();
            return (uint32_t)(acc >> (have - k)) & ((1u << k) - 1);
        }
        ORZ_HD void skip(uint32_t k) { have -= k; }
        ORZ_HD uint32_t varint(bool& bad) {  // src/coder.rs:101-113: 2 bits per payload bit, LSB first
            uint32_t v = 0;
            for (uint32_t sh = 0;; sh++) {
                const uint32_t b = bits(2);
                if (sh < 32) v |= (b & 1u) << sh;
                if (b < 2) return v;
                if (sh > 40) { bad = true; return 0; }
            }
        }
    };

    // canonical code -> lookup table (src/coder.rs:124-141, src/huffman.rs:118-167); returns max_len, 99 = bad
    ORZ_HD static uint32_t read_table(Bits& br, uint8_t* lens, uint16_t* lut, uint16_t* prim) {
        bool bad = false;
        const uint32_t max_len = br.varint(bad);
        if (!bad && max_len == 16) return 98;
        if (bad || max_len > 15) return 99;
        uint32_t ns = 0;
        for (;;) {
            const uint32_t d = br.varint(bad);
            if (bad) return 99;
            if (d == 0) break;
            if (ns + d > 512) return 99;
            for (uint32_t i = 0; i + 1 < d; i++) lens[ns++] = 0;
            const uint32_t sub = br.varint(bad);
            if (bad || sub > max_len) return 99;
            lens[ns++] = (uint8_t)(max_len - sub);
        }
        const uint32_t size = 1u << max_len;
        for (uint32_t i = 0; i < size; i++) lut[i] = 0;
        // the first-level table over the code's first P = min(max_len, 12) bits: codes of at most P bits resolve there (LDS), a
        // prefix of longer codes says so (kDecLong) and the full table in the blob is asked
        const uint32_t P = max_len < DecodeLds::kPrimBits ? max_len : DecodeLds::kPrimBits, down = max_len - P;
        for (uint32_t i = 0; i < (1u << P); i++) prim[i] = 0;
        uint32_t code = 0, cur = 1;
        for (uint32_t L = 1; L <= max_len; L++)
            for (uint32_t sy = 0; sy < ns; sy++) {
                if (lens[sy] != L) continue;
                if (L > cur) { code <<= (L - cur); cur = L; }
                const uint32_t rest = max_len - L, base = code << rest;
                if (base + (1u << rest) > size) return 99;
                const uint16_t e = (uint16_t)((sy << 4) | L);
                for (uint32_t i = 0; i < (1u << rest); i++) lut[base + i] = e;
                if (L <= P) { for (uint32_t i = 0; i < (1u << (P - L)); i++) prim[(base >> down) + i] = e; }
                else prim[base >> down] = kDecLong;
                code++;
            }
        return max_len;
    }
    ORZ_HD static uint32_t sym(Bits& br, const uint16_t* lut, const uint16_t* prim, uint32_t max_len) {
        if (max_len == 0) return 0;
        const uint32_t P = max_len < DecodeLds::kPrimBits ? max_len : DecodeLds::kPrimBits;
        uint32_t e = prim[br.peek(P)];
        if (e == kDecLong) e = lut[br.peek(max_len)];
        br.skip(e & 15);
        return e >> 4;
    }

    static size_t lds_bytes() { return DecodeLds::kBytes; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        if (w.block() >= a.count) return;
        uint32_t* l32 = (uint32_t*)w.lds();  // words[] and the ring heads start at zero (LZContext::new, src/lz.rs:57-66): all lanes clear them
        for (uint32_t i = w.lane(); i < DecodeLds::kBytes / 4; i += 64) l32[i] = 0;
        w.sync();
        l32[DecodeLds::kRecip / 4 + w.lane()] = 0xffffffffu / (327 + w.lane()) + 1;  // floor(n / d) == mulhi(n, floor(2^32 / d) + 1) for n < 2^17
        w.sync();
        if (w.lane() != 0) return;
        const uint32_t m = a.first + w.block();
        a.status[m] = run(m, a.state + (size_t)w.block() * DecodeLayout::kBytes, w.lds());
    }

    ORZ_HD uint32_t run(uint32_t m, uint8_t* st, uint8_t* lds) const {
        uint32_t* ring_pos = (uint32_t*)(st + DecodeLayout::kRingPos);
        uint8_t* ring_min = st + DecodeLayout::kRingMin;
        uint8_t* ring_exp = st + DecodeLayout::kRingExp;
        uint32_t* head = (uint32_t*)(lds + DecodeLds::kHead);
        uint16_t* prim = (uint16_t*)(lds + DecodeLds::kPrim);
        const uint32_t* recip = (const uint32_t*)(lds + DecodeLds::kRecip);
        uint16_t* rank_val = (uint16_t*)(st + DecodeLayout::kRankVal);
        uint16_t* rank_idx = (uint16_t*)(st + DecodeLayout::kRankIdx);
        uint32_t* rank_cnt = (uint32_t*)(lds + DecodeLds::kRankCnt);  // (LDS since round 6: the move targets are computed before the table's bytes arrive)
        uint32_t* rank_sum = (uint32_t*)(lds + DecodeLds::kRankSum);
        uint8_t* words = lds + DecodeLds::kWords;
        uint16_t* lut = (uint16_t*)(st + DecodeLayout::kLut);
        uint8_t* lens = st + DecodeLayout::kLens;
        uint16_t* order = (uint16_t*)(st + DecodeLayout::kOrder);
        for (uint32_t c = 0; c < 512; c++) rank_sum[c] = 1000000;  // SymRankCoder::new, src/symrank.rs:22-29

        uint8_t* out = a.out + a.out_off[m];       // out[i] is window position kPre + i
        const uint32_t out_len = a.out_len[m];
        uint64_t at = a.m_begin[m];
        const uint64_t end = a.m_end[m];
        uint32_t spos = kPre;                       // window offset, as the reference counts it
        uint32_t slid = 0;                          // bytes the window has slid by so far: window offset x is member offset x - kPre + slid
        uint64_t tail = 0;                          // the last eight decoded bytes, newest in the low byte
        bool first = true, after_literal = true;
        for (;;) {
            // chunk length, LEB128 (read_len, src/ioutil.rs:60-77); 0 = end of stream
            uint64_t t = 0;
            for (uint32_t sh = 0;; sh += 7) {
                if (at >= end || sh > 56) return kDecBadData;
                const uint8_t b = a.src[at++];
                t |= (uint64_t)(b & 0x7f) << sh;
                if (!(b & 0x80)) break;
            }
            if (t == 0) break;
            if (t >= (uint64_t)kPre * 3 || at + t > end) return kDecBadData;  // src/lib.rs:111-113
            Bits br{a.src + at, (uint32_t)t, 0, 0, 0};
            at += t;
            bool bad = false;
            if (first) {  // census order, src/lz.rs:372-392
                const uint32_t k = br.varint(bad);
                if (bad || k > kSyms) return kDecBadData;
                for (uint32_t i = 0; i < kSyms; i++) rank_idx[i] = 0xffff;  // (context 0's index array as the "seen" set)
                uint32_t n = 0;
                for (uint32_t i = 0; i < k; i++) {
                    const uint32_t v = br.bits(9);
                    if (v >= kSyms || rank_idx[v] != 0xffff) return kDecBadData;
                    rank_idx[v] = (uint16_t)n;
                    order[n++] = (uint16_t)v;
                }
                for (uint32_t v = 0; v < kSyms; v++)
                    if (rank_idx[v] == 0xffff) order[n++] = (uint16_t)v;
                for (uint32_t c = 0; c < 512; c++)
                    for (uint32_t i = 0; i < kSyms; i++) {
                        rank_val[c * kSyms + i] = order[i];
                        rank_idx[c * kSyms + order[i]] = (uint16_t)i;
                    }
                first = false;
            }
            const uint32_t end_field = br.varint(bad);
            const uint32_t n_items = br.varint(bad);
            if (bad) return kDecBadData;
            uint32_t ml[3];
            for (int k = 0; k < 3; k++) {
                ml[k] = read_table(br, lens, lut + (size_t)k * 32768, prim + (size_t)k * 4096);
                if (ml[k] == 99) return kDecBadData;
                if (ml[k] == 98) return kDecDeepTable;
            }
            for (uint32_t it = 0; it < n_items; it++) {
                const uint32_t r = sym(br, lut + (after_literal ? 32768 : 0), prim + (after_literal ? 4096 : 0), ml[after_literal ? 1 : 0]);
                if (r >= kSyms) return kDecBadData;
                // hash1(spos-1), hash2(spos-1) from the last three bytes (src/lz.rs:482-492)
                const uint8_t b1 = (uint8_t)tail, b2 = (uint8_t)(tail >> 8), b3 = (uint8_t)(tail >> 16);
                const uint32_t ctx = (uint32_t)(b1 & 0x7f) | ((uint32_t)is_alnum(b2) << 7);
                const uint32_t wkey = (uint32_t)(b1 & 0x7f) | ((((uint32_t)(b2 & 0x7f)) | ((uint32_t)is_alnum(b3) << 7)) << 7);
                const uint8_t w0 = words[wkey * 2], w1 = words[wkey * 2 + 1];
                // SymRankCoder::decode + update, src/symrank.rs:49-97
                const uint32_t c = ctx | (after_literal ? 256u : 0u);
                uint16_t* val = rank_val + (size_t)c * kSyms;
                uint16_t* idx = rank_idx + (size_t)c * kSyms;
                // The rank is r or r + 1 (or the excluded symbol's), and count and sum live in LDS: the move targets of BOTH candidates
                // are known before a byte of the table has arrived, so everything the item needs of the table -- index[excluded],
                // value[r], value[r + 1] and the two values each candidate's move displaces -- is ONE round trip (round 5: three)
                uint32_t cnt = rank_cnt[c], sum0 = rank_sum[c];
                if (cnt > kSyms) { cnt = cnt * 9 / 10; sum0 = sum0 * 9 / 10; }
                cnt += 1;
                // next_i and the mid point for rank i (src/symrank.rs:67-74)
                auto targets = [&](uint32_t i, uint32_t& ni, uint32_t& mid) {
                    const uint32_t n16 = (sum0 + i) >> 4;
                    // (sum / 16) / cnt: for the steady-state counts 327..390 by a 32-bit reciprocal -- floor(n / d) ==
                    // mulhi(n, floor(2^32 / d) + 1) for n < 2^17 (the identity orz_symrank_kernel uses; tests/test_abi.py checks it)
                    const uint32_t quo = (cnt >= 327 && cnt <= 390 && n16 < (1u << 17)) ? (uint32_t)(((uint64_t)n16 * (uint64_t)recip[cnt - 327]) >> 32) : n16 / cnt;
                    const uint32_t dec = (i / 16 + (uint16_t)quo) & 0xffff;
                    ni = i > dec ? i - dec : 0;
                    if (ni < i / 2) ni = i / 2;
                    mid = ni + (i - ni) / 2;
                };
                const uint32_t ra = r < kSyms - 1 ? r : 0, rb = r + 1 < kSyms ? r + 1 : 0;
                uint32_t nia, mida, nib, midb;
                targets(ra, nia, mida);
                targets(rb, nib, midb);
                const uint32_t iu = idx[w0];
                const uint16_t va = val[ra], vb = val[rb], xa = val[mida], ya = val[nia], xb = val[midb], yb = val[nib];
                const uint32_t i = r == kSyms - 1 ? iu : r + (r >= iu ? 1u : 0u);
                if (i >= kSyms) return kDecBadData;
                uint16_t v, x, y;
                uint32_t ni, mid;
                if (r != kSyms - 1) {
                    const bool first = i == ra;
                    v = first ? va : vb; x = first ? xa : xb; y = first ? ya : yb;
                    ni = first ? nia : nib; mid = first ? mida : midb;
                } else {  // the excluded symbol itself (rare): its rank came with the loads above
                    targets(i, ni, mid);
                    v = val[i]; x = val[mid]; y = val[ni];
                }
                rank_cnt[c] = cnt; rank_sum[c] = sum0 + i;
                if (i != ni) {  // value[i] <- value[mid] <- value[ni] <- v; a move by one has mid == ni: the swap (src/symrank.rs:75-96)
                    val[i] = x; idx[x] = (uint16_t)i;
                    if (mid != ni) { val[mid] = y; idx[y] = (uint16_t)mid; }
                    val[ni] = v; idx[v] = (uint16_t)ni;
                }
                // the last item of a stream may run past the announced end (the chunk's end field cuts it back,
                // src/lz.rs:478): such bytes are decoded but not stored -- the next member's bytes live there
                const uint32_t opos = spos - kPre + slid;
                if (opos >= out_len) return kDecSizeMismatch;  // an item STARTING past the end: not a stream of this size
                uint32_t ro = 0, len = 0, mn_raw = 0;  // (mn_raw: the source node's len_min as the match read it -- its update below needs no second load)
                bool match = false;
                if (v == kWordSym) {
                    out[opos] = w0;
                    if (opos + 1 < out_len) out[opos + 1] = w1;
                    tail = (tail << 16) | ((uint64_t)w0 << 8) | w1;
                    after_literal = false;
                } else if (v < 256) {
                    out[opos] = (uint8_t)v;
                    tail = (tail << 8) | v;
                    after_literal = true;
                } else {
                    const uint32_t roid = (v - 256) / 6, lenid = (v - 256) % 6;
                    // base(roid) = sum over k < roid of 2^(k / 2) (src/lz.rs:516-530) in closed form: j = roid / 2 full pairs give
                    // 2 (2^j - 1), an odd roid one more term 2^j
                    const uint32_t j2 = roid >> 1, base = 2 * ((1u << j2) - 1) + ((roid & 1) << j2);
                    ro = base + br.bits(roid >> 1);
                    if (ro >= kRing) return kDecBadData;
                    const uint32_t hd = head[ctx];
                    const uint32_t node = hd >= ro ? hd - ro : hd + kRing - ro;  // (head + N - ro) % N without the division: head, ro < N
                    const uint32_t enc = lenid == 5 ? sym(br, lut + 2 * 32768, prim + 2 * 4096, ml[2]) : lenid;
                    // ring nodes hold 1 + the member offset of their item (0 = never written: the reference's pos 0)
                    const uint32_t srec = ring_pos[(size_t)ctx * kRing + node];
                    uint32_t mn = ring_min[(size_t)ctx * kRing + node], ex = ring_exp[(size_t)ctx * kRing + node];
                    mn_raw = mn;
                    if (mn < kMinLen) mn = kMinLen;
                    if (ex < kMinLen) ex = kMinLen;
                    len = enc + mn > ex ? enc + mn : (enc > 0 ? enc + mn - 1 : ex);  // src/lz.rs:459-467
                    // dead: never written, or slid out of the window (window offset = member offset + kPre - slid <= 0)
                    if (srec == 0 || (uint64_t)(srec - 1) + kPre <= (uint64_t)slid || srec - 1 >= opos || len > kMaxLen + 127) return kDecBadData;
                    const uint32_t src = srec - 1;
                    const uint32_t len8 = (len + 7) & ~7u;
                    if (src + len8 <= opos && opos + len <= out_len) {
                        // the source lies wholly before the item (and the last load's spare bytes too): eight bytes a load, the loads
                        // of 32 bytes asked for together before their stores (mem_fast_copy, src/mem.rs:74-92, for a lane); the
                        // register of the last eight bytes from one more load of the source's end
                        const uint64_t endw = len >= 8 ? ldu64(out + src + len - 8) : 0;
                        uint64_t first = 0;
                        for (uint32_t k0 = 0; k0 < len; k0 += 32) {
                            uint64_t q[4];
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) q[j] = k0 + 8 * j < len ? ldu64(out + src + k0 + 8 * j) : 0;
                            if (k0 == 0) first = q[0];
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) {
                                const uint32_t k = k0 + 8 * j;
                                if (k + 8 <= len) stu64(out + opos + k, q[j]);
                                else for (uint32_t t = k; t < len; t++) out[opos + t] = (uint8_t)(q[j] >> (8 * (t - k)));
                            }
                        }
                        if (len >= 8) tail = __builtin_bswap64(endw);
                        else tail = (tail << (8 * len)) | (__builtin_bswap64(first) >> (64 - 8 * len));
                    } else {
                        for (uint32_t k = 0; k < len; k++) {  // overlap-safe forward copy
                            const uint32_t sp = src + k;
                            const uint8_t b = sp < out_len ? out[sp] : 0;
                            if (opos + k < out_len) out[opos + k] = b;
                            tail = (tail << 8) | b;
                        }
                    }
                    after_literal = false;
                    match = true;
                }
                // Bucket::update, src/matcher.rs:62-80
                {
                    const uint32_t h = head[ctx], nh = h + 1 == kRing ? 0 : h + 1;
                    if (match && len >= kMinLen) {
                        const uint32_t ni = h >= ro ? h - ro : h + kRing - ro;
                        if (mn_raw <= len) ring_min[(size_t)ctx * kRing + ni] = (uint8_t)(len + 1 < 127 ? len + 1 : 127);
                    }
                    ring_pos[(size_t)ctx * kRing + nh] = opos + 1;
                    ring_min[(size_t)ctx * kRing + nh] = 0;
                    ring_exp[(size_t)ctx * kRing + nh] = (uint8_t)(match ? len : 0);
                    head[ctx] = nh;
                }
                spos += v == kWordSym ? 2 : (match ? len : 1);
                if (v != kWordSym) {  // words[hash2(spos-3)] = the two bytes before spos (src/lz.rs:203,233)
                    const uint8_t c3 = (uint8_t)(tail >> 16), c4 = (uint8_t)(tail >> 24), c5 = (uint8_t)(tail >> 32);
                    const uint32_t k2 = (uint32_t)(c3 & 0x7f) | ((((uint32_t)(c4 & 0x7f)) | ((uint32_t)is_alnum(c5) << 7)) << 7);
                    words[k2 * 2] = (uint8_t)(tail >> 8);
                    words[k2 * 2 + 1] = (uint8_t)tail;
                }
            }
            if (end_field < spos) {  // src/lz.rs:478: an item overran the chunk's end (never in a stream of this encoder; crafted ones)
                spos = end_field;
                // the contexts of the next item come from the bytes before the CUT position -- the reference computes them from
                // the window (hash1(sbuf, spos - 1)), the register here still holds the overrun bytes
                if (spos >= kPre) {
                    const uint64_t mo = (uint64_t)spos - kPre + slid;
                    if (mo > out_len) return kDecSizeMismatch;  // (a cut that lands past the member's region -- those bytes were never stored)
                    tail = 0;
                    for (uint32_t k = 8; k >= 1; k--) tail = (tail << 8) | (mo >= k ? (uint64_t)out[mo - k] : 0);
                }
            }
            if (spos < kPre) return kDecBadData;
            if (spos >= kBlock) {  // src/lib.rs:119-124: the window slides by 2^24, every ring position with it
                if (slid > 0xffffffffu - 2 * kNewMax) return kDecBadData;  // (members below 4 GiB)
                slid += kNewMax;
                spos = kPre;
            }
        }
        return spos - kPre + slid == out_len ? kDecOk : kDecSizeMismatch;
    }
};


// ------------------------------------------------------------------------------------------------ host side
struct DecodeStats {
    uint64_t members = 0, in_bytes = 0, out_bytes = 0, launches = 0;
    double kernel_ms = 0, total_s = 0;
};

struct MemberIndex {  // the container cut into members, from the chunk framing alone (no decoding)
    std::vector<uint64_t> begin, end, out_off;
    std::vector<uint32_t> out_len;
    uint64_t out_total = 0;
};

// Walks the chunk framing of every member (LEB128 lengths, src/ioutil.rs:60-77) and reads each chunk's end
// field from its prologue (first chunk of a member: after the census, src/lz.rs:372-395): that is the
// member's decoded size (a chunk that ends at the block's end slides the window, src/lib.rs:119-124: the next chunk's
// field counts from SBVEC_PREMATCH_LEN again).  Throws std::runtime_error on malformed framing.
inline MemberIndex index_members(const uint8_t* src, size_t n) {
    MemberIndex ix;
    size_t at = 0;
    while (at < n) {
        const size_t begin = at;
        uint32_t spos_end = kPre;
        uint64_t slid = 0;
        bool first = true;
        for (;;) {
            uint64_t t = 0;
            for (unsigned sh = 0;; sh += 7) {
                if (at >= n || sh > 56) throw std::runtime_error("invalid orz data: truncated chunk length");
                const uint8_t b = src[at++];
                t |= (uint64_t)(b & 0x7f) << sh;
                if (!(b & 0x80)) break;
            }
            if (t == 0) break;
            if (t >= (uint64_t)kPre * 3 || at + t > n) throw std::runtime_error("invalid orz data: chunk length");
            DecodeMember::Bits br{src + at, (uint32_t)t, 0, 0, 0};
            bool bad = false;
            if (first) {
                const uint32_t k = br.varint(bad);
                if (bad || k > kSyms) throw std::runtime_error("invalid orz data: census");
                for (uint32_t i = 0; i < k; i++) br.bits(9);
                first = false;
            }
            const uint32_t end_field = br.varint(bad);
            if (bad || end_field < spos_end || end_field > kBlock) throw std::runtime_error("invalid orz data: end field");
            spos_end = end_field;
            if (spos_end >= kBlock) {  // the decoder slides here (src/lib.rs:119-124)
                slid += kNewMax;
                spos_end = kPre;
                if (slid > 0xffffffffull - 2 * kNewMax) throw std::runtime_error("member of 4 GiB or more: use the host decoder");
            }
            at += t;
        }
        // A member cannot code more than 4096 bytes per byte of its own: an item is at least one bit behind a non-empty
        // Huffman table and at most 255 + 127 bytes long.  Streams that announce more (tables of zero-length codes,
        // which no encoder writes, or plain lies about the size) are not sized on the device on their own word.
        const uint64_t mlen = (uint64_t)(spos_end - kPre) + slid;
        if (mlen > (uint64_t)(at - begin) * 4096 + 4096)
            throw std::runtime_error("member announces more output than its bits can code: use the host decoder");
        ix.begin.push_back(begin);
        ix.end.push_back(at);
        ix.out_off.push_back(ix.out_total);
        ix.out_len.push_back((uint32_t)mlen);
        ix.out_total += mlen;
    }
    return ix;
}

// Decodes every member of the container on the backend's device; `slots` members are in flight at once
// (one wavefront and one 7.3 MB state blob each).
template <class BE>
void decode_members_device(BE& be, const uint8_t* src, size_t n, std::vector<uint8_t>& out, DecodeStats& stats,
                           uint32_t slots = 2048) {
    const double t0 = be.now();
    const MemberIndex ix = index_members(src, n);
    const uint32_t M = (uint32_t)ix.begin.size();
    out.assign(ix.out_total, 0);
    stats.members = M; stats.in_bytes = n; stats.out_bytes = ix.out_total;
    if (M == 0) { stats.total_s = be.now() - t0; return; }
    if (slots > M) slots = M;
    uint8_t* d_src = be.template alloc<uint8_t>(n);
    uint8_t* d_out = be.template alloc<uint8_t>(ix.out_total);
    uint64_t* d_begin = be.template alloc<uint64_t>(M);
    uint64_t* d_end = be.template alloc<uint64_t>(M);
    uint64_t* d_off = be.template alloc<uint64_t>(M);
    uint32_t* d_len = be.template alloc<uint32_t>(M);
    uint32_t* d_status = be.template alloc<uint32_t>(M);
    uint8_t* d_state = be.template alloc<uint8_t>((size_t)slots * DecodeLayout::kBytes);
    be.h2d(d_src, src, n);
    be.h2d(d_begin, ix.begin.data(), (size_t)M * 8);
    be.h2d(d_end, ix.end.data(), (size_t)M * 8);
    be.h2d(d_off, ix.out_off.data(), (size_t)M * 8);
    be.h2d(d_len, ix.out_len.data(), (size_t)M * 4);
    be.set_timing(true);
    uint64_t dummy = 0;
    be.collect_timed(&dummy);
    for (uint32_t first = 0; first < M; first += slots) {
        const uint32_t count = M - first < slots ? M - first : slots;
        if (first) be.memset(d_state, 0, (size_t)slots * DecodeLayout::kBytes);  // (alloc zeroes the first round)
        be.timed_begin(2);  // (a slot that is recorded without profile mode)
        be.launch_waves(count, DecodeMember{DecodeArgs{d_src, d_begin, d_end, d_off, d_len, d_out, d_state, d_status, first, count}}, DecodeMember::lds_bytes());
        be.timed_end(2);
        stats.launches++;
    }
    uint64_t nl = 0, nby[4];
    double msby[4];
    be.collect_timed(&nl, msby, nby);
    stats.kernel_ms = msby[2];
    be.set_timing(false);
    std::vector<uint32_t> status(M);
    be.d2h(status.data(), d_status, (size_t)M * 4);
    be.d2h(out.data(), d_out, ix.out_total);
    for (void* p : {(void*)d_src, (void*)d_out, (void*)d_begin, (void*)d_end, (void*)d_off, (void*)d_len, (void*)d_status, (void*)d_state}) be.free(p);
    for (uint32_t m = 0; m < M; m++)
        if (status[m] != kDecOk)
            throw std::runtime_error(status[m] == kDecTooLarge ? "member larger than one block: use the host decoder"
                                     : status[m] == kDecDeepTable ? "member with a 16-bit Huffman table: use the host decoder"
                                                               : "invalid orz data (member " + std::to_string(m) + ", status " + std::to_string(status[m]) + ")");
    stats.total_s = be.now() - t0;
}

}  // namespace orz
