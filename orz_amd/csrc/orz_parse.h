// orz_parse.h -- the ROLZ parse of one stream block as speculative, wave-cooperative sweeps.
//
// What this replaces in the reference (/root/reference, Rust):
//   LZEncoder::encode parse loop            src/lz.rs:131-235
//   BucketMatcher::find_match / has_lazy    src/matcher.rs:135-228   (+ LCP, src/mem.rs:41-70)
//   Bucket::update / BucketMatcher::update  src/matcher.rs:62-80,115-121
//   words[] predictor                       src/lz.rs:132-136,203,233
// The reference runs them as one serial chain over the stream.  Here the block is cut into
// segments of <= 62 positions; one wavefront owns one segment and re-derives its items with the
// reference's exact decision rules, reading every other segment's items as they currently stand
// (speculation).  A sweep evaluates a window of segments beyond the `front` (first non-final
// segment); segments that sit before the first segment whose result changed are final, because
// everything they read was final.  The fixed point is the serial parse.
//
// Data layout (one block, HBM):
//   static, built by the prep kernels once per block
//     srec[slot].pos  positions sorted by (ctx8, hash_dword % 4627, position)  ("candidate lists")
//     idx[x]          slot of window offset x
//     runstart[key]   first slot of a (ctx8, hash) run
//     kpos/kidx/krun  the same for the word predictor, keyed by hash2 (15 bit)
//   speculative, updated in place by the owning segment's wavefront (only when a value changes)
//     vbits           1 bit per slot: position is an item start (ring member)
//     srec[slot].ml   the item's match_len_expected (0 literal/word, 255 = not an item)
//     srec[slot].ord  the item's ordinal in its ctx ring (recomputed per sweep by the rank kernel)
//     kbits           1 bit per word-predictor slot: words[] was updated at that position + 2
//     exitst[s]       how segment s-1 passed the stream on: the entry it walked from and the exit it reached
//                     ((next item position << 2) | last type, each), a `settled` flag and the sweep that wrote
//                     it -- see ExitPair and the hand-off inside a sweep in phase 2
//     hist[s][ctx]    items per context of segment s (ring of R segments) -> base[s][ctx] prefix
#pragma once
#include "orz_common.h"

namespace orz {

constexpr uint32_t kSegMax = 62;            // positions per segment; + 2 lazy probe positions = 64 lanes
constexpr uint32_t kNPMax = kSegMax + 2;
constexpr uint32_t kNoChange = 0xffffffffu;
constexpr uint32_t kRankChunk = 32;         // segments per rank-kernel block
constexpr uint32_t kLbPre = 8;              // bytes staged in LDS before the segment start
constexpr uint32_t kLbLen = 336;            // kLbPre + 64 + 240 + 8 rounded up to dwords: x+240+8 readable
constexpr uint32_t kRecBatch = 8;           // slot records in flight per lane (8 x 32 B: register budget)
constexpr uint32_t kCntSlack = 64;          // own items a segment can add to one context (robustness margin)

struct SlotRec {  // one candidate-list slot: 32 bytes, half a cache line, fetched as one request
    uint32_t pos;
    uint32_t ord;
    uint32_t ml;
    uint32_t pad;
    uint64_t t0, t1;  // the 16 text bytes at pos (static): most common-prefix lengths are settled without
                      // touching the window
};

// How a segment passed the stream on, packed into one 64-bit word that is stored and loaded atomically:
//   [26:0] entry = (position of the first item at or after the segment start << 2) | type of the item before
//   [35:27] exit position - entry position (< 512: a segment is <= 62 bytes, an item <= 240)
//   [37:36] type of the last item   [38] settled (will not change any more in this sweep)   [63:39] sweep
struct ExitPair {
    ORZ_HD static uint64_t make(uint32_t sweep, bool settled, uint32_t entry, uint32_t exit) {
        return ((uint64_t)sweep << 39) | ((uint64_t)(settled ? 1 : 0) << 38) | ((uint64_t)(exit & 3) << 36) |
               ((uint64_t)(((exit >> 2) - (entry >> 2)) & 0x1ff) << 27) | (entry & 0x7ffffffu);
    }
    ORZ_HD static uint32_t entry(uint64_t e) { return (uint32_t)e & 0x7ffffffu; }
    ORZ_HD static uint32_t exit(uint64_t e) {
        return ((((uint32_t)e & 0x7ffffffu) >> 2) + (uint32_t)((e >> 27) & 0x1ff)) << 2 | (uint32_t)((e >> 36) & 3);
    }
    ORZ_HD static bool settled(uint64_t e) { return (e >> 38) & 1; }
    ORZ_HD static uint32_t sweep(uint64_t e) { return (uint32_t)(e >> 39); }
};

struct ParseCtl {           // device-resident sweep control of one stream
    uint32_t front[2];      // first non-final segment, by sweep parity
    uint32_t fchg[2];       // lowest segment whose result changed in the sweep, by parity
    uint32_t evals;         // segments evaluated so far (statistics)
    uint32_t nprof;         // waves sampled into prof[]
    uint32_t slow;          // items that needed the serial evaluation (statistics)
    uint32_t wend;          // end of the last sweep's window: segments >= wend were never evaluated
    uint32_t skipped;       // evaluations given up because the wave ran late (statistics)
    uint32_t pad0;
    unsigned long long prof[8];  // shader cycles per phase, summed over the sampled waves
    unsigned long long prof2[8]; // phase 1 detail: max-over-lanes stamps
    unsigned long long prof3[8]; // the same for the slowest waves (phase 1 > 140 K cycles); [6] = their number
    uint32_t stop_cause[32];     // per sweep: input-change bits of the first changed segment (2 entry, 4 words, 8 candidates, 16 first evaluation)
    uint32_t cause[8];           // changed segments near the front: [0] all, [1] entry moved, [2] words answers moved,
                                 // [3] candidate lists moved, [4] none of these
    uint32_t p1_hist[16];        // sampled waves: cycles until the end of phase 1, 16 K per bucket
    uint32_t adv_hist[16];       // histogram of the front's advance per sweep: bucket = floor(log2(segments + 1))
};

struct ParseArgs {
    const uint8_t* win;       // window offset 0 (480 readable bytes before it)
    uint32_t len;             // kPre + n
    uint32_t nseg, seg;       // segments in the block, positions per segment (<= 62)
    uint32_t wsegs;           // segments per sweep window
    uint32_t ring;            // R: rows of the hist/base rings (>= wsegs + 1)
    uint32_t depth, lazy1, lazy2, dmax;  // LZCfg, dmax = max of the three = candidates kept per position
    uint32_t lt0;             // type of the last item of the previous block (after_literal carry)
    uint32_t par;             // sweep parity
    uint32_t prof;            // sample phase timings into ctl->prof (diagnostics)
    uint32_t sweep;           // 1-based id of this launch within the block (stamps the exit states)
    uint32_t polls;           // hand-off: poll cap per wait (the wall-clock deadline is what normally ends a wait)
    uint32_t chain;           // hand-off look-back: a wave watches the pairs of this many predecessors (<= 63; 1 = no
                              // hand-off inside a sweep)
    SlotRec* srec;
    const uint32_t* idx;
    const uint32_t* runstart;
    const uint32_t* kpos;
    const uint32_t* kidx;
    const uint32_t* krun;
    const uint8_t* wsnap;     // words[] as of the block start, [32768][2]
    uint64_t* vbits;          // level 0: one bit per slot
    uint64_t* v1;             // level 1: one bit per vbits word that may be non-zero (set-only inside a block)
    uint64_t* v2;             // level 2: one bit per v1 word that may be non-zero
    uint64_t* kbits;          // word-predictor slots, same three levels
    uint64_t* k1;
    uint64_t* k2;
    uint64_t* exitst;         // [nseg + 2]  ExitPair of segment s-1 at [s]
    uint32_t maxpass = 3;     // walks again at most this many times per sweep
    uint32_t deadline = 0;    // stop waiting for hand-offs this long after the wave started (10 ns ticks; 0 = polls only)
    uint32_t near = 0;        // blocks >= near (far from the front) use the two limits below instead
    uint32_t far_deadline = 0;
    uint32_t skip_after = 0;  // a far wave still collecting candidates this long after its start keeps its previous
                              // evaluation instead (0 = never)
    uint32_t skip_rand = 0;   // test hook: skip pseudo-randomly, about one evaluation in skip_rand
    uint8_t* hist;            // [ring][256]
    uint32_t* base;           // [ring][256]
    uint8_t* TY;              // per position outputs of the owning segment
    uint32_t* SRC;
    uint8_t* W0;
    uint8_t* LR;
    uint32_t* sig;            // [nseg][4] diagnostics (ORZ_PROF): input signatures of each segment's last evaluation
    unsigned long long* tim = nullptr;  // [wsegs][8] diagnostics (ORZ_TIMELINE): wall-clock stamps of one sweep's waves
    uint32_t timsweep = 0;    // the sweep `tim` records
    uint32_t* partial;        // [2][chunks][256] per-chunk ctx item counts of the sweep, by parity
    ParseCtl* ctl;
};

#if defined(__HIP_DEVICE_COMPILE__)
// Unaligned loads.  The alignment has to sit on a TYPEDEF: `reinterpret_cast<const __attribute__((aligned(1))) uint64_t*>` is accepted and the
// attribute silently dropped -- the compiler then takes the address for 8-aligned, which the vector memory path forgives but the
// SCALAR path does not: with a wave-uniform address the load becomes an s_load and returns other bytes.  (Found in round 5: every
// use until then had a per-lane address; PathMarkWave's one uniform load wrote the bytes of 48 positions earlier.)
typedef uint32_t orz_u32_unaligned __attribute__((aligned(1)));
typedef uint64_t orz_u64_unaligned __attribute__((aligned(1)));
ORZ_D uint32_t ldu32(const uint8_t* p) { return *reinterpret_cast<const orz_u32_unaligned*>(p); }
ORZ_D uint64_t ldu64(const uint8_t* p) { return *reinterpret_cast<const orz_u64_unaligned*>(p); }
ORZ_D void stu64(uint8_t* p, uint64_t v) { *reinterpret_cast<orz_u64_unaligned*>(p) = v; }
ORZ_D void atom_or64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
ORZ_D void atom_and64(uint64_t* p, uint64_t v) { atomicAnd((unsigned long long*)p, (unsigned long long)v); }
ORZ_D uint64_t atom_xchg64(uint64_t* p, uint64_t v) { return atomicExch((unsigned long long*)p, (unsigned long long)v); }
ORZ_D uint64_t atom_load64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ORZ_D void atom_store64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ORZ_D void spin_pause() { __builtin_amdgcn_s_sleep(2); }
ORZ_D void atom_min32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
ORZ_D void atom_max32(uint32_t* p, uint32_t v) { atomicMax(p, v); }
ORZ_D void atom_add32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
ORZ_D void atom_add64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
ORZ_D int clz64(uint64_t v) { return __clzll((long long)v); }
ORZ_D int ctz64(uint64_t v) { return __ffsll((long long)v) - 1; }
ORZ_D SlotRec ld_rec(const SlotRec* p) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[0];
    const uint4 t = reinterpret_cast<const uint4*>(p)[1];
    return SlotRec{v.x, v.y, v.z, v.w, (uint64_t)t.x | ((uint64_t)t.y << 32), (uint64_t)t.z | ((uint64_t)t.w << 32)};
}
#else
ORZ_D uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
ORZ_D uint64_t ldu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
ORZ_D void stu64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
// (real relaxed atomics on the host as well: the race check of tests/race runs a launch's threads on several host threads)
ORZ_D void atom_or64(uint64_t* p, uint64_t v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
ORZ_D void atom_and64(uint64_t* p, uint64_t v) { __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
ORZ_D uint64_t atom_xchg64(uint64_t* p, uint64_t v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
ORZ_D uint64_t atom_load64(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
ORZ_D void atom_store64(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
ORZ_D void spin_pause() {}
ORZ_D void atom_min32(uint32_t* p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
ORZ_D void atom_max32(uint32_t* p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
ORZ_D void atom_add32(uint32_t* p, uint32_t v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
ORZ_D void atom_add64(unsigned long long* p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
ORZ_D int clz64(uint64_t v) { return __builtin_clzll(v); }
ORZ_D int ctz64(uint64_t v) { return __builtin_ctzll(v); }
ORZ_D SlotRec ld_rec(const SlotRec* p) { return *p; }
#endif

// common prefix of a[0..) and b[0..), capped, eight bytes a step (== src/mem.rs:41-51)
ORZ_D uint32_t lcp240u(const uint8_t* a, const uint8_t* b, uint32_t cap = kMaxLen) {
    uint32_t l = 0;
    while (l < cap) {
        uint64_t x = ldu64(a + l) ^ ldu64(b + l);
        if (x) {
            l += (uint32_t)ctz64(x) >> 3;
            return l < cap ? l : cap;
        }
        l += 8;
    }
    return cap;
}

// Three-level bitmap over the candidate slots: L0 (bit per slot), L1 (bit per non-zero L0 word),
// L2 (bit per non-zero L1 word).  ParseWave updates level 0 only (plain atomics, nothing to wait
// for); the rank kernel rebuilds both summary levels from level 0 after every sweep
// (`rebuild_summaries`), so they are exact whenever a sweep starts; inside a sweep setters also set the
// summary bits (never clear them).  A walker that misses a bit set during its own sweep has merely
// speculated; the setter reports the change.
ORZ_D void slot_set(uint64_t* L0, uint64_t* L1, uint64_t* L2, uint32_t j) {
    const uint32_t w = j >> 6;
    atom_or64(&L0[w], 1ull << (j & 63));
    atom_or64(&L1[w >> 6], 1ull << (w & 63));  // fire-and-forget: lets walkers of this very sweep find the bit
    atom_or64(&L2[w >> 12], 1ull << ((w >> 6) & 63));
}
ORZ_D void slot_clear(uint64_t* L0, uint32_t j) { atom_and64(&L0[j >> 6], ~(1ull << (j & 63))); }

// One block (256 threads) per level-2 word = 4096 level-0 words: writes its 64 level-1 words and its
// level-2 word.  `sh` = 64 u64 of LDS; sync() = block barrier; t = thread id.
template <class SYNC>
ORZ_D void rebuild_summaries(const uint64_t* L0, uint64_t* L1, uint64_t* L2, uint32_t nwords0, uint32_t blk, uint32_t t,
                             uint64_t* sh, SYNC sync) {
    // thread t owns level-1 bits [16 t', ...): 16 consecutive level-0 words -> a quarter of level-1 word t/4
    const uint32_t w0 = blk * 4096 + t * 16;
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = w0 + i < nwords0 ? L0[w0 + i] : 0;
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) bits |= (uint32_t)(v[i] != 0) << i;
    uint16_t* piece = (uint16_t*)sh;  // [256] sixteen level-1 bits per thread
    piece[t] = (uint16_t)bits;
    sync();
    // (only the level-1 words that summarise existing level-0 words: the last block of a full 16 MiB block's word-list bitmap
    // -- 262,146 words, 65 blocks -- used to write 62 words past the level-1 array, zeros, harmless behind a buffer of its own
    // and fatal beside a neighbour: with the encoder's buffers side by side they wiped the level-2 summary and the exact
    // parse missed word updates.  Found in round 4 with ORZ_ARENA_MB / ORZ_EMU_ARENA_MB.)
    if (t < 64 && (size_t)(blk * 64 + t) * 64 < nwords0)
        L1[blk * 64 + t] = (uint64_t)piece[4 * t] | ((uint64_t)piece[4 * t + 1] << 16) | ((uint64_t)piece[4 * t + 2] << 32) |
                           ((uint64_t)piece[4 * t + 3] << 48);
    if (t == 64) {
        uint64_t m = 0;
        for (int i = 0; i < 64; i++)
            m |= (uint64_t)((piece[4 * i] | piece[4 * i + 1] | piece[4 * i + 2] | piece[4 * i + 3]) != 0) << i;
        L2[blk] = m;
    }
}

// Newest-first list of the set slots in [lo, hi) of a three-level bitmap, at most D of them.
// `word` is the already loaded level-0 word of slot hi-1.  Older words are reached through the
// summaries: only non-empty ones are loaded, four in flight a round.
template <class Late>
ORZ_D uint32_t collect_slots(const uint64_t* L0, const uint64_t* L1, const uint64_t* L2, uint64_t word, uint64_t l1word,
                             uint32_t hi, uint32_t lo, uint32_t D, uint32_t* out, uint32_t ostride, uint32_t& nwords,
                             const Late& late) {
    uint32_t found = 0;
    if (hi <= lo) return 0;
    const uint32_t w0 = (hi - 1) >> 6, wmin = lo >> 6;
    if (hi & 63) word &= (1ull << (hi & 63)) - 1;
    if ((w0 << 6) < lo) word &= ~0ull << (lo - (w0 << 6));
    while (word && found < D) {
        int bit = 63 - clz64(word);
        out[ostride * found++] = (w0 << 6) + (uint32_t)bit;
        word &= ~(1ull << bit);
    }
    uint32_t u = w0 >> 6;
    bool more = found < D && w0 > wmin;
    uint64_t m1 = more ? l1word : 0;  // level-1 word of slot hi-1, loaded together with `word`
    if (w0 & 63) m1 &= (1ull << (w0 & 63)) - 1; else m1 = 0;
    while (more) {
        if ((u << 6) < wmin) m1 &= ~0ull << (wmin - (u << 6));
        uint32_t round = 0;
        while (m1 && found < D) {
            // four words in flight in the first round, eight from then on (sparse runs: fewer round trips)
            uint32_t ww[8];
            uint64_t wd[8];
            const int cap = round++ ? 8 : 4;
            int nw = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ww[i] = 0;
                if (i < cap && m1) {
                    const int bit = 63 - clz64(m1);
                    m1 &= ~(1ull << bit);
                    ww[i] = (u << 6) + (uint32_t)bit;
                    nw = i + 1;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) wd[i] = i < nw ? L0[ww[i]] : 0;
            nwords += (uint32_t)nw;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint64_t v = wd[i];
                if (i < nw && (ww[i] << 6) < lo) v &= ~0ull << (lo - (ww[i] << 6));
                while (v && found < D) {
                    int bit = 63 - clz64(v);
                    out[ostride * found++] = (ww[i] << 6) + (uint32_t)bit;
                    v &= ~(1ull << bit);
                }
            }
        }
        if (found >= D || (u << 6) <= wmin || u == 0 || late()) break;
        uint32_t u2 = (u - 1) >> 6;  // next non-empty level-1 word below u, through level 2
        uint64_t m2 = L2[u2];
        if (u & 63) m2 &= (1ull << (u & 63)) - 1;
        for (;;) {
            if (m2) { u = (u2 << 6) + (uint32_t)(63 - clz64(m2)); break; }
            if (u2 == 0 || (u2 << 12) <= wmin) { more = false; break; }
            u2--;
            m2 = L2[u2];
        }
        if (!more || ((u << 6) + 63) < wmin) break;
        m1 = L1[u];
        nwords++;
    }
    return found;
}

// decision record of one position (u64): what the item starting there would be
constexpr uint64_t kDecFull = 1ull << 39, kDecBl = 1ull << 38, kDecMatch = 1ull << 33, kDecLazy1 = 1ull << 34, kDecLazy2a = 1ull << 35, kDecLazy2b = 1ull << 36,
                   kDecRobust = 1ull << 37;
ORZ_D uint32_t dec_src(uint64_t d) { return (uint32_t)(d & 0x1ffffff); }
ORZ_D uint32_t dec_len(uint64_t d) { return (uint32_t)(d >> 25) & 0xff; }

// byte offsets of the per-wave LDS arrays
struct ParseLds {
    uint32_t lb, cdat, dec, idxL, keyL, kkL, ownord, srcL, basec, mlz;
    uint32_t wg, ctxL, ncand, ownv, ownml, ownE, oldml, oldE, tyL, w0L, lrL, cnt, dbg, total;
    ORZ_HD static ParseLds make(uint32_t dmax, bool prof = false) {
        ParseLds o;
        uint32_t at = 0;
        auto take = [&](uint32_t bytes) { uint32_t r = at; at += (bytes + 15) & ~15u; return r; };
        o.lb = take(kLbLen + 16);
        o.cdat = take(kNPMax * dmax * 8);  // per candidate: ro0 (16) | lcp << 16 | ml << 24 | pos << 32
        o.dec = take(kNPMax * 8);
        o.idxL = take(kNPMax * 4);
        o.keyL = take(kNPMax * 4);
        o.kkL = take((kNPMax + 2) * 4);
        o.ownord = take(kNPMax * 4);
        o.srcL = take(kNPMax * 4);
        o.basec = take(256 * 4);
        o.mlz = take(kNPMax * 4);          // per position: what its list offers a lazy probe (see precompute)
        o.wg = take(kNPMax * 2);
        o.ctxL = take(kNPMax);
        o.ncand = take(kNPMax);
        o.ownv = take(kNPMax);
        o.ownml = take(kNPMax);
        o.ownE = take(kNPMax);
        o.oldml = take(kNPMax);
        o.oldE = take(kNPMax);
        o.tyL = take(kNPMax);
        o.w0L = take(kNPMax);
        o.lrL = take(kNPMax);
        o.cnt = take(256);
        o.dbg = take(prof ? 64 * 8 * 4 : 0);  // profiling stamps of sampled waves
        o.total = at;
        return o;
    }
};

// ---------------------------------------------------------------------------------------------
// One wavefront = one segment.  W is the wave context: lane(), block(), lds(), ballot(), sync().
struct ParseWave {
    ParseArgs a;

    // pointers into the wave's LDS
    struct Sh {
        uint8_t* lb;
        uint64_t* cdat;
        uint64_t* dec;
        uint32_t *idxL, *keyL, *kkL, *ownord, *srcL, *basec, *mlz;
        uint16_t* wg;
        uint8_t *ctxL, *ncand, *ownv, *ownml, *ownE, *oldml, *oldE, *tyL, *w0L, *lrL, *cnt;
    };

    template <class W>
    ORZ_D void operator()(W& w) const {
        const uint32_t lane = w.lane();
        const uint32_t front = a.ctl->front[a.par];
        const uint32_t sg = front + w.block();
        if (sg >= a.nseg) return;
        const ParseLds L = ParseLds::make(a.dmax, a.prof & 1);
        uint8_t* lds = w.lds();
        Sh s;
        s.lb = lds + L.lb;  // lb[kLbPre + i] = win[seg_start + i]
        s.cdat = (uint64_t*)(lds + L.cdat);
        s.dec = (uint64_t*)(lds + L.dec);
        s.idxL = (uint32_t*)(lds + L.idxL);
        s.keyL = (uint32_t*)(lds + L.keyL);
        s.kkL = (uint32_t*)(lds + L.kkL);  // kkL[i] = hash2(seg_start - 2 + i - 1)
        s.ownord = (uint32_t*)(lds + L.ownord);
        s.srcL = (uint32_t*)(lds + L.srcL);
        s.basec = (uint32_t*)(lds + L.basec);
        s.mlz = (uint32_t*)(lds + L.mlz);
        s.wg = (uint16_t*)(lds + L.wg);
        s.ctxL = lds + L.ctxL; s.ncand = lds + L.ncand; s.ownv = lds + L.ownv; s.ownml = lds + L.ownml;
        s.ownE = lds + L.ownE; s.oldml = lds + L.oldml; s.oldE = lds + L.oldE; s.tyL = lds + L.tyL;
        s.w0L = lds + L.w0L; s.lrL = lds + L.lrL; s.cnt = lds + L.cnt;

        const uint8_t* b = a.win;
        const uint32_t seg_start = kPre + sg * a.seg;
        const uint32_t seg_end = seg_start + a.seg < a.len ? seg_start + a.seg : a.len;
        const uint32_t npos = seg_end - seg_start;  // item-start positions of the segment (<= 62)
        const uint32_t nprobe = npos + 2;           // + lazy probe positions (<= 64 = one per lane)
        const uint32_t D = a.dmax;
        const bool prof = (a.prof & 1) && (w.block() & 63) == 5;
        unsigned long long tk0 = prof ? w.clock() : 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
        const bool tl = a.tim && a.sweep == a.timsweep;  // timeline of this sweep (diagnostics)
        const unsigned long long tl0 = w.wallclock();
        unsigned long long tl1 = 0, tl2 = 0, tl3 = 0;
        uint32_t tlpolls = 0, tlpass = 0;

        // ---- phase 0: stage the segment's bytes, slots, old state and the ctx ordinals in LDS
        const uint64_t oldpair = w.bcast64(lane == 0 ? atom_load64(&a.exitst[sg + 1]) : 0, 0);  // my last evaluation's hand-off
        // a wave far from the front that runs late may keep its previous evaluation (if it has one)
        const bool far = a.near && w.block() >= a.near;
        const bool may_skip = far && ExitPair::sweep(oldpair) != 0;
        const bool skip_now = may_skip && a.skip_rand && ((sg * 2654435761u) ^ (a.sweep * 40503u)) % a.skip_rand == 0;
        bool gave_up = false;
        auto late = [&]() -> bool {
            if (skip_now || (may_skip && a.skip_after && w.wallclock() - tl0 > a.skip_after)) gave_up = true;
            return gave_up;
        };
        {
            const uint8_t* src = b + (int64_t)seg_start - kLbPre;
            for (uint32_t i = lane * 4; i < kLbLen; i += 256) {
                uint32_t v = ldu32(src + i);
                __builtin_memcpy(s.lb + i, &v, 4);
            }
            const uint32_t x = lane;
            const uint32_t pos = seg_start + x;
            uint32_t j = 0;
            if (x < nprobe && pos < a.len) j = a.idx[pos];
            uint32_t ks = 0;
            const bool hasE = x < npos && pos >= kPre + 1;
            if (hasE) ks = a.kidx[pos - 2];
            // a segment that enters the window for the first time has no ordinal row of its own yet: the row
            // of the previous window end is a far better guess than a stale ring row
            const uint32_t wprev = a.ctl->wend;
            const uint32_t brow = ((a.prof & 2) == 0 && sg > wprev ? wprev : sg) % a.ring;
            for (uint32_t c = lane; c < 256; c += 64) {
                s.basec[c] = a.base[(size_t)brow * 256 + c];
                s.cnt[c] = 0;
            }
            if (x < nprobe) {
                s.idxL[x] = j;
                s.ncand[x] = 0;
                s.dec[x] = 0;
            }
            if (x < npos) {
                s.oldml[x] = (uint8_t)a.srec[j].ml;
                s.oldE[x] = hasE ? (uint8_t)((a.kbits[ks >> 6] >> (ks & 63)) & 1) : 0;
                s.ownv[x] = 0; s.ownml[x] = 0; s.ownE[x] = 0;
            }
        }
        w.sync();
        if (lane < nprobe) {
            const uint32_t x = lane;
            const uint8_t* px = s.lb + kLbPre + x;
            uint32_t c = (uint32_t)(px[-1] & 0x7f) | ((uint32_t)is_alnum(px[-2]) << 7);  // hash1(pos-1)
            s.ctxL[x] = (uint8_t)c;
            s.keyL[x] = c * kHash + hash_entry(px);
        }
        for (uint32_t i = lane; i < nprobe + 2; i += 64) {  // position u = seg_start - 2 + i: hash2(u - 1)
            const uint8_t* pu = s.lb + kLbPre - 2 + i;
            uint32_t h1 = (uint32_t)(pu[-2] & 0x7f) | ((uint32_t)is_alnum(pu[-3]) << 7);
            s.kkL[i] = (uint32_t)(pu[-1] & 0x7f) | (h1 << 7);
        }
        w.sync();
        if (prof) tk1 = w.clock();

        // ---- phase 1: every position of the segment collects, on its own lane, the candidates that
        // older segments offer it: the most recent <= D ring members of its (ctx, hash) run, their
        // ordinals, expected lengths and common-prefix lengths; and the word predictor's answer.
        // how many of this segment's earlier positions share my (ctx, hash) key / my words[] key:
        // their slots sit right below mine in the run and are this sweep's business, not older segments'
        uint32_t same = 0, samek = 0;
        {   // which lanes hold the same key as mine?  one ballot per key bit (21 + 15) instead of 64 broadcasts
            const uint32_t kmine = lane < nprobe ? s.keyL[lane] : 0x1fffffu;  // probe lanes beyond the segment: no real key
            const uint32_t kkmine = s.kkL[lane];                               // key of entry u = seg_start - 2 + lane
            uint64_t mk = ~0ull, mkk = ~0ull;
            for (uint32_t bit = 0; bit < 21; bit++) {
                const uint64_t bal = w.ballot((kmine >> bit) & 1);
                mk &= ((kmine >> bit) & 1) ? bal : ~bal;
            }
            for (uint32_t bit = 0; bit < 15; bit++) {
                const uint64_t bal = w.ballot((kkmine >> bit) & 1);
                mkk &= ((kkmine >> bit) & 1) ? bal : ~bal;
            }
            const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0;          // lanes < mine
            const uint64_t real = nprobe >= 64 ? ~0ull : ((1ull << nprobe) - 1);  // lanes that hold a position
            same = (uint32_t)__builtin_popcountll(mk & below & real);
            // my words[] lookup key is the entry key of lane + 2: take that lane's match mask
            const uint32_t src = lane + 2 < 64 ? lane + 2 : lane;
            const uint64_t mq = (uint64_t)w.shfl((uint32_t)mkk, src) | ((uint64_t)w.shfl((uint32_t)(mkk >> 32), src) << 32);
            const uint32_t ulo = seg_start - 2 >= kPre - 1 ? 0 : (kPre - 1) - (seg_start - 2);  // entries u >= P-1 only
            const uint64_t upto = lane + 2 >= 64 ? ~0ull : ((1ull << (lane + 2)) - 1);          // entries i < lane + 2
            samek = (uint32_t)__builtin_popcountll(mq & upto & (~0ull << ulo));
        }
        if (lane < nprobe && seg_start + lane < a.len) {
            const uint32_t x = lane;
            const uint32_t pos = seg_start + x;
            const uint32_t key = s.keyL[x];
            // slots of this segment's own positions sit right below idx[x] in the run (same key, ascending
            // position): they are this sweep's business, not the older segments'
            const bool wantw = x < npos;
            const uint32_t kk = s.kkL[x + 2];
            const uint32_t hi = s.idxL[x] - same;
            uint32_t* dbg = (uint32_t*)(lds + L.dbg) + lane * 8;
            if (prof) dbg[0] = (uint32_t)(w.clock() - tk1);
            const uint32_t khi = wantw ? a.kidx[pos] - samek : 0;
            // first round trip: run starts and the first bitmap words of both lists
            const uint32_t lo = a.runstart[key];
            const uint32_t klo = wantw ? a.krun[kk] : 0;
            uint64_t word = hi ? a.vbits[(hi - 1) >> 6] : 0;
            const uint64_t l1word = hi ? a.v1[(hi - 1) >> 12] : 0;
            uint64_t kword = (wantw && khi) ? a.kbits[(khi - 1) >> 6] : 0;
            const uint64_t kl1word = (wantw && khi) ? a.k1[(khi - 1) >> 12] : 0;
            uint32_t wsn = wantw ? (uint32_t)a.wsnap[kk * 2] | ((uint32_t)a.wsnap[kk * 2 + 1] << 8) : 0;

            uint32_t found = 0, nwords = 0;
            uint64_t* mydat = s.cdat + x * D;
            uint32_t* mycq = (uint32_t*)mydat;  // slot k waits in the low half of entry k until its record arrives
            if (prof) dbg[1] = (uint32_t)(w.clock() - tk1) + (uint32_t)(word & 0) + (uint32_t)(kword & 0) + (uint32_t)(l1word & 0) + (uint32_t)(kl1word & 0) + (lo & 0) + (klo & 0) + (wsn & 0);
            if (late()) found = 0;
            else found = collect_slots(a.vbits, a.v1, a.v2, word, l1word, hi, lo, D, mycq, 2, nwords, late);
            if (prof) dbg[2] = (uint32_t)(w.clock() - tk1);
            uint32_t kslot = 0xffffffffu;
            if (wantw && !gave_up && collect_slots(a.kbits, a.k1, a.k2, kword, kl1word, khi, klo, 1, &kslot, 1, nwords, late) == 0) kslot = 0xffffffffu;
            if (prof) { dbg[3] = (uint32_t)(w.clock() - tk1); dbg[7] = nwords; }
            // second round trip: the slot records (32 B each, text included), sixteen in flight
            const uint32_t ku = kslot != 0xffffffffu ? a.kpos[kslot] : 0;
            const uint8_t* px = s.lb + kLbPre + x;
            const uint64_t x0 = ldu64(px), x1 = ldu64(px + 8);
            const uint32_t hc0 = s.basec[s.ctxL[x]];  // ring ordinal of an item starting here, before own items
            Pre pre;
            for (uint32_t k0 = 0; k0 < found && !late(); k0 += kRecBatch) {
                SlotRec r[kRecBatch];
                uint32_t l[kRecBatch];
                uint32_t act = 0;  // candidates whose common prefix is still growing
#pragma unroll
                for (int i = 0; i < (int)kRecBatch; i++) r[i] = ld_rec(&a.srec[mycq[2 * (k0 + i < found ? k0 + i : k0)]]);
#pragma unroll
                for (int i = 0; i < (int)kRecBatch; i++) {
                    const uint64_t d0 = r[i].t0 ^ x0, d1 = r[i].t1 ^ x1;
                    if (d0) l[i] = (uint32_t)ctz64(d0) >> 3;
                    else if (d1) l[i] = 8 + ((uint32_t)ctz64(d1) >> 3);
                    else { l[i] = 16; if (k0 + i < found && r[i].ml != 255) act |= 1u << i; }
                }
                // the long ones advance together (all at the same offset), 64 bytes a round; the position's
                // own bytes are read from LDS once per round
                for (uint32_t off = 16; act && off < kMaxLen; off += 64) {
                    uint64_t xb[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) xb[k] = ldu64(px + off + 8 * k);
#pragma unroll
                    for (int i = 0; i < (int)kRecBatch; i++) {
                        if (act & (1u << i)) {
                            const uint8_t* qa = b + r[i].pos + off;
                            uint64_t d[8];
#pragma unroll
                            for (int k = 0; k < 8; k++) d[k] = ldu64(qa + 8 * k) ^ xb[k];
                            uint32_t adv = 64;
#pragma unroll
                            for (int k = 7; k >= 0; k--) if (d[k]) adv = 8 * (uint32_t)k + ((uint32_t)ctz64(d[k]) >> 3);
                            l[i] = off + adv;
                            if (adv < 64) act &= ~(1u << i);
                            if (l[i] >= kMaxLen) { l[i] = kMaxLen; act &= ~(1u << i); }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < (int)kRecBatch; i++) {
                    if (k0 + i < found) {
                        const uint32_t ll = r[i].ml == 255 ? 0 : l[i];
                        const uint32_t ro = hc0 - 1 - r[i].ord;  // reduced offset if no own item intervenes
                        const uint32_t ro16 = ro < 0xffffu ? ro : 0xffffu;  // saturated: far outside the 4094-ring
                        mydat[k0 + i] = (uint64_t)ro16 | ((uint64_t)ll << 16) | ((uint64_t)(r[i].ml & 0xff) << 24) |
                                        ((uint64_t)r[i].pos << 32);
                        pre_feed(pre, b, px, ro16, ll, r[i].ml & 0xff, r[i].pos);
                    }
                }
            }
            if (prof) dbg[4] = (uint32_t)(w.clock() - tk1);
            s.ncand[x] = (uint8_t)found;
            pre_finish(pre, s, seg_start, x, x < npos);
            if (wantw) {
                if (kslot != 0xffffffffu) wsn = (uint32_t)b[ku] | ((uint32_t)b[ku + 1] << 8);
                s.wg[x] = (uint16_t)wsn;
            }
        }
        w.sync();
        // running late, far from the front: the segment keeps its previous evaluation.  Hand the old exit on under
        // this sweep's stamp, count the old items into the ordinal prefix (the hist row is still ours), and hold
        // the front back (the segment was not evaluated against this sweep's state).
        auto keep_previous = [&]() {
            if (lane == 0) {
                atom_store64(&a.exitst[sg + 1], ExitPair::make(a.sweep, true, ExitPair::entry(oldpair), ExitPair::exit(oldpair)));
                atom_add32(&a.ctl->skipped, 1);
                atom_min32(&a.ctl->fchg[a.par], sg - 1);
            }
            uint32_t* part = a.partial + ((size_t)a.par * (a.wsegs / kRankChunk + 1) + w.block() / kRankChunk) * 256;
            for (uint32_t c = lane; c < 256; c += 64) {
                const uint32_t v = a.hist[(size_t)(sg % a.ring) * 256 + c];
                if (v) atom_add32(&part[c], v);
            }
        };
        if (w.ballot(gave_up)) { keep_previous(); return; }
        if (prof) tk2 = w.clock();
        uint32_t sigW = 0, sigC = 0;
        if ((a.prof & 1) && a.sig) {  // diagnostics: what did this evaluation read? (compared with the last one at the end)
            uint32_t hw = 0, hc = 0;
            if (lane < npos) hw = (uint32_t)s.wg[lane] * 2654435761u + lane;
            if (lane < nprobe && seg_start + lane < a.len)
                for (uint32_t k = 0; k < s.ncand[lane]; k++) {
                    const uint64_t cd = s.cdat[lane * D + k];
                    hc = hc * 31u + (uint32_t)(cd >> 32) + (uint32_t)((cd >> 16) & 0xffff) * 977u;  // pos, lcp, ml (not the ring offset)
                }
            for (int off = 32; off; off >>= 1) { hw ^= w.shfl(hw, lane ^ off); hc += w.shfl(hc, lane ^ off); }
            sigW = hw; sigC = hc;
        }

        // ---- phase 2: the segment's items.  Everything up to here did not depend on the entry state (where the
        // previous segment left the stream).  The wave walks optimistically from the entry it knows and publishes
        // the pair (entry, exit); it then watches the pairs its `chain` predecessors publish in THIS sweep: the
        // predecessor left the stream elsewhere -> walk again from there; every link back to a settled segment
        // agrees -> settled.  A parse that shifts and re-synchronises a segment or two later thus settles within
        // one sweep instead of costing one sweep per segment.
        const bool anchor = sg == 0 || a.chain <= 1 || w.block() == 0;  // the front segment's entry is exact
        const uint32_t dl = far ? a.far_deadline : a.deadline;
        const uint32_t look = anchor ? 0 : (w.block() < a.chain ? w.block() : (a.chain < 63 ? a.chain : 63));
        uint32_t ventry = 0, vold = 0;
        if (sg == 0) ventry = (kPre << 2) | a.lt0;
        {   // lanes 1..5: what the up to five segments before this one last reported (a long match may have
            // skipped some of them)
            uint32_t v = 0;
            if (lane >= 1 && lane <= 5 && lane <= sg) v = ExitPair::exit(atom_load64(&a.exitst[sg + 1 - lane]));
            vold = ExitPair::exit(oldpair);
            for (int off = 4; off; off >>= 1) { const uint32_t o = w.shfl(v, lane ^ off); if (o > v) v = o; }
            if (sg != 0) ventry = w.bcast(v, 0);
        }
        const uint32_t mykey = lane < npos ? s.keyL[lane] : 0xffffffffu;
        const uint32_t mykk = lane < npos ? s.kkL[lane] : 0xffffffffu;
        uint32_t p = 0, lt = 0, nslow = 0;
        bool exit_changed = false;
        if (prof) tk3 = w.clock();
        if (tl) tl1 = w.wallclock();
        for (uint32_t pass = 0;; pass++) {
            tlpass = pass;
            if (pass) {  // walking again: forget the first attempt
                for (uint32_t c = lane; c < 256; c += 64) s.cnt[c] = 0;
                if (lane < npos) { s.ownv[lane] = 0; s.ownml[lane] = 0; s.ownE[lane] = 0; }
            }
            p = ventry >> 2;
            lt = ventry & 3;
            // each lane keeps its own position's flags in registers: item start / words[] update there
            bool myv = false, myE = false;
            if (sg != 0 && p < seg_end) {
                if (lane == p - seg_start) myE = (lt != kTyWord);
                if (lane == 0) s.ownE[p - seg_start] = (lt != kTyWord);
            }
            w.sync();
            // ---- phase 2b: walk the items.  Robust decisions are taken as they are; the others are
            // evaluated on the spot with this sweep's own items as the most recent candidates.
                while (p < seg_end) {
                const uint32_t x = p - seg_start;
                const bool older = lane < x && myv;
                const uint64_t m0 = w.ballot(older && mykey == s.keyL[x]);
                const uint64_t m1 = w.ballot(older && mykey == s.keyL[x + 1]);
                const uint64_t m2 = w.ballot(older && mykey == s.keyL[x + 2]);
                const uint64_t mE = w.ballot(lane <= x && myE && mykk == s.kkL[x + 2]);
                const uint8_t* px = s.lb + kLbPre + x;
                uint32_t w0, w1;
                if (mE) {  // words[hash2(p-1)] was last written by this segment: at position e = seg_start + y
                    const uint32_t y = 63 - (uint32_t)clz64(mE);
                    w0 = s.lb[kLbPre + y - 2];
                    w1 = s.lb[kLbPre + y - 1];
                } else {
                    w0 = s.wg[x] & 0xff;
                    w1 = s.wg[x] >> 8;
                }
                const uint32_t lwm = (px[0] == w0 && px[1] == w1);
                uint64_t d = s.dec[x];
                bool is_match = (d & kDecMatch) != 0;
                uint32_t max_len = dec_len(d), lazy = 0;
                bool robust = (d & kDecRobust) != 0 && !m0;
                if (is_match && max_len < kMaxLen / 2) {  // the lazy probes read what the lists of x+1 / x+2 offer
                    const uint32_t l1 = max_len + 1 + ((d & kDecBl) ? 1u : 0u);
                    const uint32_t z1 = s.mlz[x + 1], z2 = s.mlz[x + 2];
                    if ((z1 & 0xff) >= l1) lazy = 1;
                    else if (((z2 >> 8) & 0xff) >= l1 - lwm) lazy = 2;
                    robust = robust && ((z1 >> 16) & 1) && ((z2 >> 17) & 1) && !(m1 | m2);
                }
    #ifdef ORZ_FORCE_SLOW
                robust = false;
    #endif
                if (!robust) {  // this sweep's own items interfere (or a ring threshold is near): evaluate on the spot
                    if (lane == 0) s.dec[x] = eval_item(s, b, seg_start, x, m0, m1, m2, lwm, false);
                    w.sync();
                    d = s.dec[x];
                    nslow++;
                    is_match = (d & kDecMatch) != 0;
                    max_len = dec_len(d);
                    lazy = (d & kDecLazy1) ? 1 : ((d & kDecLazy2a) ? 2 : 0);
                }
                // commit, src/lz.rs:172-234
                const uint32_t c = s.ctxL[x];
                const uint32_t cc = s.cnt[c];
                const uint8_t al = (lt == kTyLit) ? 4 : 0;
                uint32_t np, ty, ml = 0;
                if (is_match && !lazy) {
                    ty = kTyMatch; ml = max_len; np = p + max_len;
                } else if (p + 1 < a.len && lazy != 1 && lwm) {
                    ty = kTyWord; np = p + 2;
                } else {
                    ty = kTyLit; np = p + 1;
                }
                if (lane == 0) {
                    s.ownv[x] = 1;
                    s.ownord[x] = s.basec[c] + cc;
                    s.lrL[x] = (uint8_t)cc;
                    s.cnt[c] = (uint8_t)(cc + 1);
                    s.w0L[x] = (uint8_t)w0;
                    s.tyL[x] = (uint8_t)(ty | al);
                    s.ownml[x] = (uint8_t)ml;
                    s.srcL[x] = dec_src(d);
                    if (np < seg_end) s.ownE[np - seg_start] = (ty != kTyWord);
                }
                if (lane == x) myv = true;
                if (np < seg_end && lane == np - seg_start) myE = (ty != kTyWord);
                p = np;
                lt = ty;
                w.sync();
            }
            // ---- hand-off: publish (entry, exit) of this walk, then look at what the `look` segments before
            // this one published in THIS sweep.  The predecessor left the stream somewhere else: walk again
            // from there.  Every link back to a settled segment (or all `look` links) agrees: settled.
            if (pass == 0 && may_skip && !skip_now && a.skip_after && w.wallclock() - tl0 > a.skip_after + a.skip_after / 4) {
                keep_previous();  // a slow walk on top of it all: nothing of this evaluation is out yet
                return;
            }
            const uint32_t vexit = (p << 2) | lt;
            bool fin = anchor || pass >= a.maxpass;
            if (lane == 0) atom_store64(&a.exitst[sg + 1], ExitPair::make(a.sweep, fin, ventry, vexit));
            if (tl && pass == 0) tl2 = w.wallclock();
            uint32_t again = 0, vnew = ventry;
            for (uint32_t tries = 0; !fin; tries++) {
                // lane i (1..look) holds the pair of segment sg - i; lane 0 stands for this segment
                uint64_t q = 0;
                if (lane >= 1 && lane <= look) q = atom_load64(&a.exitst[sg + 1 - lane]);
                const bool pub = lane >= 1 && lane <= look && ExitPair::sweep(q) == a.sweep;
                const uint32_t myE = lane == 0 ? ventry : ExitPair::entry(q);
                const uint32_t succE = w.shfl(myE, lane ? lane - 1 : 0);     // entry of the segment after mine
                const uint64_t P = w.ballot(pub) | 1;                         // bit 0: this segment
                const bool succ_pub = lane >= 1 && ((P >> (lane - 1)) & 1);
                const uint64_t C = w.ballot(pub && succ_pub && ExitPair::exit(q) == succE);
                const uint64_t F = w.ballot(pub && ExitPair::settled(q));
                const uint32_t x1 = w.bcast(ExitPair::exit(q), 1);
                const bool overdue = dl && w.wallclock() - tl0 > dl;
                if (!overdue && (P & 2) && !(C & 2)) {  // the predecessor left the stream elsewhere: follow at once, even if it
                                                        // may move again (waiting for it to settle first measured 6 % slower)
                    again = 1; vnew = x1;
                    break;
                }
                const uint64_t upto = F ? ((2ull << ctz64(F)) - 2) : ((2ull << look) - 2);  // links 1 .. nearest settled
                if ((C & upto) == upto) break;                                // settled
                if (tries >= a.polls || overdue) break;  // give up: later sweeps sort it out
                spin_pause();
                tlpolls++;
            }
            if (again) { ventry = vnew; continue; }
            if (!fin && lane == 0) atom_store64(&a.exitst[sg + 1], ExitPair::make(a.sweep, true, ventry, vexit));
            exit_changed = vexit != vold;
            break;
        }
        if (prof) tk4 = w.clock();
        if (tl) tl3 = w.wallclock();

        // ---- phase 3: publish what changed, the per-ctx item counts and the exit state
        bool changed = false;
        if (lane < npos) {
            const uint32_t x = lane;
            const uint32_t pos = seg_start + x;
            const uint32_t j = s.idxL[x];
            const uint32_t nm = s.ownv[x] ? s.ownml[x] : 255u;
            const uint32_t om = s.oldml[x];
            if (nm != om) {
                a.srec[j].ml = nm;
                if ((nm == 255) != (om == 255)) {
                    if (nm == 255) slot_clear(a.vbits, j);
                    else slot_set(a.vbits, a.v1, a.v2, j);
                }
                changed = true;
            }
            if (s.ownv[x]) {
                a.TY[pos] = s.tyL[x];
                a.W0[pos] = s.w0L[x];
                a.LR[pos] = s.lrL[x];
                if ((s.tyL[x] & 3) == kTyMatch) a.SRC[pos] = s.srcL[x];
            }
            if (pos >= kPre + 1 && s.ownE[x] != s.oldE[x]) {
                const uint32_t ks = a.kidx[pos - 2];
                if (s.ownE[x]) slot_set(a.kbits, a.k1, a.k2, ks);
                else slot_clear(a.kbits, ks);
                changed = true;
            }
        }
        {
            uint32_t* part = a.partial + ((size_t)a.par * (a.wsegs / kRankChunk + 1) + w.block() / kRankChunk) * 256;
            for (uint32_t c = lane; c < 256; c += 64) {
                const uint32_t v = s.cnt[c];
                a.hist[(size_t)(sg % a.ring) * 256 + c] = (uint8_t)v;
                if (v) atom_add32(&part[c], v);
            }
        }
        if (lane == 0) {
            if (exit_changed) changed = true;
            atom_add32(&a.ctl->evals, 1);
            if (nslow) atom_add32(&a.ctl->slow, nslow);
        }
        const bool anych = w.ballot(changed) != 0;
        if ((a.prof & 1) && a.sig && lane == 0) {
            uint32_t* sg4 = a.sig + (size_t)sg * 4;
            if (anych && w.block() < 256 && (sg4[3] & 1)) {
                const bool e = sg4[0] != ventry, ww = sg4[1] != sigW, c = sg4[2] != sigC;
                atom_add32(&a.ctl->cause[0], 1);
                if (e) atom_add32(&a.ctl->cause[1], 1);
                if (ww) atom_add32(&a.ctl->cause[2], 1);
                if (c) atom_add32(&a.ctl->cause[3], 1);
                if (!e && !ww && !c) atom_add32(&a.ctl->cause[4], 1);
            }
            const uint32_t bits = sg4[3] ? ((sg4[0] != ventry ? 2u : 0u) | (sg4[1] != sigW ? 4u : 0u) | (sg4[2] != sigC ? 8u : 0u)) : 16u;
            sg4[0] = ventry; sg4[1] = sigW; sg4[2] = sigC; sg4[3] = 1 | (bits << 8);
        }
        if (anych && lane == 0) atom_min32(&a.ctl->fchg[a.par], sg);
        if (tl && lane == 0) {
            unsigned long long* t = a.tim + (size_t)w.block() * 8;
            t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = tl3; t[4] = w.wallclock();
            t[5] = ((unsigned long long)tlpass << 32) | tlpolls; t[6] = sg; t[7] = anych;
        }
        if (prof && lane == 0) {
            const unsigned long long tk5 = w.clock();
            atom_add64(&a.ctl->prof[0], tk1 - tk0);
            atom_add64(&a.ctl->prof[1], tk2 - tk1);
            atom_add64(&a.ctl->prof[2], tk3 - tk2);
            atom_add64(&a.ctl->prof[3], tk4 - tk3);
            atom_add64(&a.ctl->prof[4], tk5 - tk4);
            atom_add32(&a.ctl->nprof, 1);
            {   // distribution of the wave's time up to the end of phase 1 (shader cycles): buckets of 16 K
                const unsigned long long t1 = tk2 - tk0;
                uint32_t bk = (uint32_t)(t1 >> 14);
                if (bk > 15) bk = 15;
                atom_add32(&a.ctl->p1_hist[bk], 1);
            }
            {
                const uint32_t* dg = (const uint32_t*)(lds + L.dbg);
                const bool slowwave = (tk2 - tk0) > 140000;  // tail waves are accounted separately (prof3)
                for (int k = 0; k < 5; k++) {
                    uint32_t mx = 0;
                    for (uint32_t i = 0; i < nprobe && seg_start + i < a.len; i++) if (dg[i * 8 + k] > mx) mx = dg[i * 8 + k];
                    atom_add64(slowwave ? &a.ctl->prof3[k] : &a.ctl->prof2[k], mx);
                }
                if (slowwave) atom_add64(&a.ctl->prof3[6], 1);
                uint32_t mw = 0;
                for (uint32_t i = 0; i < nprobe && seg_start + i < a.len; i++) if (dg[i * 8 + 7] > mw) mw = dg[i * 8 + 7];
                atom_add64(&a.ctl->prof2[7], mw);
            }
        }
    }

    // One pass of a position over its own candidate list, on its own lane, assuming none of the segment's
    // own items interfere: (a) find_match for an item starting here -> dec[x]; (b) what the list offers a
    // lazy probe AT this position: the longest common prefix among its first lazy1 / lazy2 ring members
    // (has_lazy_match(min_len) <=> that maximum >= min_len) -> mlz[x].  Robust bits: the result does not
    // change for any own-item count in [0, kCntSlack].  Candidates are fed newest first, straight from
    // the registers of phase 1.
    struct Pre {
        uint32_t max_len = kMinLen - 1, mlexp = kMinLen, bestq = 0, bestro = 0, nmain = 0, n1 = 0, n2 = 0, M1 = 0, M2 = 0;
        bool rmain = true, r1 = true, r2 = true, stop_main = false, stop_all = false;
    };
    ORZ_D void pre_feed(Pre& q, const uint8_t* b, const uint8_t* px, uint32_t ro16, uint32_t l, uint32_t ml, uint32_t pos) const {
        if (q.stop_all || ml == 255) return;
        const uint32_t ro = ro16 == 0xffff ? 0x7fffffffu : ro16;
        const bool near = ro <= kRing - 1 && ro + kCntSlack > kRing - 1;
        const bool want_main = !q.stop_main && q.nmain < a.depth, want1 = q.n1 < a.lazy1, want2 = q.n2 < a.lazy2;
        if (near) { if (want_main) q.rmain = false; if (want1) q.r1 = false; if (want2) q.r2 = false; }
        if (ro > kRing - 1) { q.stop_all = true; return; }
        if (want1) { q.n1++; if (l > q.M1) q.M1 = l; }
        if (want2) { q.n2++; if (l > q.M2) q.M2 = l; }
        if (want_main) {
            q.nmain++;
            if (l > q.max_len) {
                q.mlexp = ml; q.max_len = l; q.bestq = pos; q.bestro = ro;
                if (l == kMaxLen || (q.mlexp > 0 && l > q.mlexp)) q.stop_main = true;
            } else if (l + 3 < q.max_len && q.mlexp > 0 && l > q.mlexp) {
                // the reference's 4-byte prefilter can pass by chance past the mismatch; it then leaves the walk
                // without a better match (src/matcher.rs:150-168)
                if (ldu32(b + pos + q.max_len - 3) == ldu32(px + q.max_len - 3)) q.stop_main = true;
            }
        } else {
            q.stop_main = true;
        }
        if (q.stop_main && !want1 && !want2) q.stop_all = true;
    }
    ORZ_D void pre_finish(Pre& q, const Sh& s, uint32_t seg_start, uint32_t x, bool item_start) const {
        s.mlz[x] = q.M1 | (q.M2 << 8) | ((uint32_t)q.r1 << 16) | ((uint32_t)q.r2 << 17);
        if (!item_start) return;
        const bool is_match = q.max_len >= kMinLen && seg_start + x + q.max_len < a.len;
        uint64_t d = (uint64_t)q.bestq | ((uint64_t)q.max_len << 25);
        if (is_match) {
            d |= kDecMatch;
            const bool bl = roid_bitlen(q.bestro) < 8;
            if (bl != (roid_bitlen(q.bestro + kCntSlack) < 8)) q.rmain = false;
            if (bl) d |= kDecBl;
        }
        if (q.rmain) d |= kDecRobust;
        s.dec[x] = d;
    }

    // What the item starting at segment position x is (src/lz.rs:131-235 for one spos):
    // find_match over this sweep's own items (masks m0/m1/m2, newest first) and then the list
    // collected in phase 1, followed by the two lazy probes.  `lwm`: 0/1 = last_word_matched is
    // known; 2 = unknown, both lazy2 variants are computed.  With `check` the decision also reports
    // whether it stays the same for every own-item count in [0, kCntSlack] (kDecRobust).
    ORZ_D uint64_t eval_item(const Sh& s, const uint8_t* b, uint32_t seg_start, uint32_t x, uint64_t m0, uint64_t m1,
                             uint64_t m2, uint32_t lwm, bool check) const {
        const uint32_t D = a.dmax;
        const uint8_t* px = s.lb + kLbPre + x;
        const uint32_t p = seg_start + x;
        const uint32_t c = s.ctxL[x];
        const uint32_t hcnt = s.basec[c] + s.cnt[c];
        bool robust = true;
        // find_match, src/matcher.rs:135-192
        const uint32_t cntc = s.cnt[c];
        uint32_t max_len = kMinLen - 1, mlexp = kMinLen, bestq = 0, bestro = 0, cntv = 0;
        bool stop = false;
        for (uint64_t m = m0; m && !stop;) {
            const uint32_t y = 63 - (uint32_t)clz64(m);
            m &= ~(1ull << y);
            const uint32_t oq = s.ownord[y];
            if (hcnt - 1 - oq > kRing - 1 || cntv >= a.depth) { stop = true; break; }
            cntv++;
            const uint32_t l = lcp240u(s.lb + kLbPre + y, px);
            if (l > max_len) {
                mlexp = s.ownml[y]; max_len = l; bestq = seg_start + y; bestro = hcnt - 1 - oq;
                if (l == kMaxLen || (mlexp > 0 && l > mlexp)) stop = true;
            } else if (l + 3 < max_len && mlexp > 0 && l > mlexp) {
                if (ldu32(s.lb + kLbPre + y + max_len - 3) == ldu32(px + max_len - 3)) stop = true;
            }
        }
        const uint32_t nc = s.ncand[x];
        const uint64_t* dat = s.cdat + x * D;
        for (uint32_t k0 = 0; k0 < nc && !stop; k0 += 16) {
            uint64_t cdv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) cdv[i] = k0 + i < nc ? dat[k0 + i] : (255ull << 24);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint64_t cd = cdv[i];
                const uint32_t ml = (uint32_t)(cd >> 24) & 0xff;
                if (stop || ml == 255) continue;
                const uint32_t ro0 = (uint32_t)cd & 0xffff;
                const uint32_t ro = ro0 == 0xffff ? 0x7fffffffu : ro0 + cntc;
                if (check && ro <= kRing - 1 && ro + kCntSlack > kRing - 1) robust = false;
                if (ro > kRing - 1 || cntv >= a.depth) { stop = true; continue; }
                cntv++;
                const uint32_t l = (uint32_t)(cd >> 16) & 0xff;
                if (l > max_len) {
                    mlexp = ml; max_len = l; bestq = (uint32_t)(cd >> 32); bestro = ro;
                    if (l == kMaxLen || (mlexp > 0 && l > mlexp)) stop = true;
                } else if (l + 3 < max_len && mlexp > 0 && l > mlexp) {
                    // the reference's 4-byte prefilter can pass by chance past the mismatch; it then
                    // leaves the walk without a better match (src/matcher.rs:150-168)
                    if (ldu32(b + (uint32_t)(cd >> 32) + max_len - 3) == ldu32(px + max_len - 3)) stop = true;
                }
            }
        }
        const bool is_match = max_len >= kMinLen && p + max_len < a.len;
        uint64_t d = (uint64_t)bestq | ((uint64_t)max_len << 25);
        if (is_match) d |= kDecMatch;
        if (is_match && max_len < kMaxLen / 2) {  // src/lz.rs:151-170
            if (check && (roid_bitlen(bestro) < 8) != (roid_bitlen(bestro + kCntSlack) < 8)) robust = false;
            const uint32_t l1 = max_len + 1 + (roid_bitlen(bestro) < 8);
            if (has_lazy(s, m1, x + 1, l1, a.lazy1, check, robust)) {
                d |= kDecLazy1;
            } else if (lwm == 2) {
                if (has_lazy(s, m2, x + 2, l1, a.lazy2, check, robust)) d |= kDecLazy2a | kDecLazy2b;
                else if (has_lazy(s, m2, x + 2, l1 - 1, a.lazy2, check, robust)) d |= kDecLazy2b;
            } else if (has_lazy(s, m2, x + 2, l1 - lwm, a.lazy2, check, robust)) {
                d |= kDecLazy2a | kDecLazy2b;
            }
        }
        if (robust && check) d |= kDecRobust;  // an on-the-spot evaluation is never reused as a robust one
        return d | kDecFull;
    }

    // has_lazy_match (src/matcher.rs:194-228) for probe position xx in {x+1, x+2}: candidates are the
    // ring members inserted before p, newest first: this sweep's own items (mask), then the older
    // segments' list collected in phase 1.
    ORZ_D bool has_lazy(const Sh& s, uint64_t mown, uint32_t xx, uint32_t min_len, uint32_t depth, bool check,
                        bool& robust) const {
        const uint32_t D = a.dmax;
        const uint32_t cx = s.ctxL[xx];
        const uint32_t hx = s.basec[cx] + s.cnt[cx];
        uint32_t cntv = 0;
        for (uint64_t m = mown; m;) {
            const uint32_t y = 63 - (uint32_t)clz64(m);
            m &= ~(1ull << y);
            if (hx - 1 - s.ownord[y] > kRing - 1 || cntv >= depth) return false;
            cntv++;
            if (lcp240u(s.lb + kLbPre + y, s.lb + kLbPre + xx) >= min_len) return true;
        }
        const uint64_t* dat = s.cdat + xx * D;
        const uint32_t nc = s.ncand[xx];
        for (uint32_t k0 = 0; k0 < nc; k0 += 16) {
            uint64_t cdv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) cdv[i] = k0 + i < nc ? dat[k0 + i] : (255ull << 24);
            int res = -1;  // -1 undecided, 0 false, 1 true
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint64_t cd = cdv[i];
                if (res >= 0 || ((uint32_t)(cd >> 24) & 0xff) == 255) continue;
                const uint32_t ro0 = (uint32_t)cd & 0xffff;
                const uint32_t ro = ro0 == 0xffff ? 0x7fffffffu : ro0 + s.cnt[cx];
                if (check && ro <= kRing - 1 && ro + kCntSlack > kRing - 1) robust = false;
                if (ro > kRing - 1 || cntv >= depth) { res = 0; continue; }
                cntv++;
                if (((uint32_t)(cd >> 16) & 0xff) >= min_len) res = 1;
            }
            if (res >= 0) return res == 1;
        }
        return false;
    }
};

// ---------------------------------------------------------------------------------------------
// After a sweep: ring ordinals.  One block of 256 threads (thread = ctx c) per chunk of
// kRankChunk segments of the window:
//   off[c]       = base[front][c] + sum of the partial[chunk' < chunk][c] that ParseWave accumulated
//   base[s+1][c] = base[s][c] + hist[s][c] over the chunk's segments            (ring of R rows)
//   srec.ord     = base[seg][ctx] + LR for every current item of the chunk
// Block 0 also moves the front to just past the first changed segment and re-arms the other parity.
struct RankArgs {
    const uint8_t* win;
    ParseCtl* ctl;
    const uint8_t* hist;
    uint32_t* base;
    uint32_t* partial;
    const uint32_t* idx;
    SlotRec* srec;
    const uint8_t* LR;
    uint32_t nseg, seg, wsegs, ring, len, par;
    const uint64_t *vbits, *kbits;  // level-0 bitmaps ...
    uint64_t *v1, *v2, *k1, *k2;     // ... and their summaries, rebuilt here after every sweep
    uint32_t nvwords, nkwords;       // level-0 words in use
    const uint32_t* sig;             // diagnostics (may be null)
};
// `rows` = LDS [kRankChunk + 1][256] u32 ; sync() = block barrier ; c = thread id (0..255)
template <class SYNC>
ORZ_D void rank_chunk(const RankArgs& a, uint32_t chunk, uint32_t c, uint32_t* rows, SYNC sync) {
    const uint32_t f = a.ctl->front[a.par];
    const uint32_t wend = f + a.wsegs < a.nseg ? f + a.wsegs : a.nseg;
    const uint32_t nchunk = a.wsegs / kRankChunk + 1;
    const uint32_t s0 = f + chunk * kRankChunk;
    const uint32_t s1 = s0 + kRankChunk < wend ? s0 + kRankChunk : wend;
    const uint32_t* part = a.partial + (size_t)a.par * nchunk * 256;
    const bool live = s0 < wend;
    // everything below is arranged as three rounds of independent loads: (1) ordinal inputs and the
    // chunk's positions, (2) their slot records, then the barrier, then only stores
    const uint32_t x0 = kPre + s0 * a.seg;
    const uint32_t npos = live ? (s1 - s0) * a.seg : 0;  // <= 32 * 62 = 1984 <= 8 positions per thread
    const uint64_t segmagic = 0x100000000ull / a.seg + 1;
    uint32_t j[8], ml[8], lr[8], cx[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t i = c + (uint32_t)k * 256, x = x0 + i;
        ok[k] = i < npos && x < a.len;
        j[k] = ok[k] ? a.idx[x] : 0;
        lr[k] = ok[k] ? a.LR[x] : 0;
        cx[k] = ok[k] ? hash1(a.win, x - 1) : 0;
    }
    if (live) {
        uint32_t run = a.base[(size_t)(f % a.ring) * 256 + c];
        uint32_t h[kRankChunk];
        const uint32_t r0 = s0 % a.ring;  // ring rows of the chunk: r0, r0+1, ... wrapping once at most
#pragma unroll
        for (uint32_t i = 0; i < kRankChunk; i++) {
            uint32_t r = r0 + i;
            if (r >= a.ring) r -= a.ring;
            h[i] = s0 + i < s1 ? a.hist[(size_t)r * 256 + c] : 0;
        }
        for (uint32_t i0 = 0; i0 < chunk; i0 += 64) {  // sums of the chunks before mine, 64 loads in flight
            uint32_t v[64];
#pragma unroll
            for (int k = 0; k < 64; k++) v[k] = i0 + k < chunk ? part[(i0 + k) * 256 + c] : 0;
#pragma unroll
            for (int k = 0; k < 64; k++) run += v[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) ml[k] = ok[k] ? a.srec[j[k]].ml : 255u;
        rows[c] = run;
#pragma unroll
        for (uint32_t i = 0; i < kRankChunk; i++) {
            if (s0 + i < s1) {
                uint32_t r = r0 + i + 1;
                if (r >= a.ring) r -= a.ring;
                run += h[i];
                rows[(i + 1) * 256 + c] = run;
                a.base[(size_t)r * 256 + c] = run;
            }
        }
    }
    sync();
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = c + (uint32_t)k * 256;
            const uint32_t sgi = (uint32_t)(((uint64_t)i * segmagic) >> 32);  // == i / seg for i < 2^16
            if (ok[k] && ml[k] != 255) a.srec[j[k]].ord = rows[sgi * 256 + cx[k]] + lr[k];
        }
    }
    // re-arm the other parity's accumulators for the next sweep (nobody reads them in this launch)
    uint32_t* other = a.partial + (size_t)(a.par ^ 1) * nchunk * 256;
    other[chunk * 256 + c] = 0;
    if (chunk == 0 && c == 0) {
        const uint32_t fc = a.ctl->fchg[a.par];
        uint32_t nf = fc == kNoChange ? wend : fc + 1;
        if (f >= a.nseg) nf = f;
        a.ctl->front[a.par ^ 1] = nf;
        a.ctl->fchg[a.par ^ 1] = kNoChange;
        if (wend > a.ctl->wend) a.ctl->wend = wend;
        if (a.sig && fc != kNoChange) {  // diagnostics: what had moved for the segment that stopped the front
            const uint32_t bits = (a.sig[(size_t)fc * 4 + 3] >> 8) & 31;
            a.ctl->stop_cause[bits & 31]++;
        }
        {
            uint32_t adv = nf - f, bk = 0;
            while ((2u << bk) <= adv + 1 && bk < 15) bk++;
            if (f < a.nseg) a.ctl->adv_hist[bk]++;
        }
    }
}

}  // namespace orz
