// orz_fast.h -- the GPU-native ("fast") parse mode of the MI355X ROLZ encoder.
//
// The reference parse (LZEncoder::encode, /root/reference/src/lz.rs:131-235) is one serial chain: the
// ring of a context holds item starts only (src/matcher.rs:62-80), so which positions may serve as match
// sources is decided by the parse itself.  The decoder, however, accepts ANY parse its state machine can
// express.  The fast mode keeps the reference's decision rules but evaluates them for
// every position at once against a snapshot of the item starts, and removes the serial chain with a
// pipelined Gauss-Seidel schedule:
//
//   * the block's new bytes are cut into tiles of `tile` positions; at step s tile t runs its round s - t
//     (1..R): all active tiles re-decide every position from the current snapshot, then the path through
//     the active range is re-extracted.  A tile's last round therefore sees every earlier tile final, so
//     only sources inside its own tile can have been stale.
//   * candidate lists are static per block: positions radix-sorted by (ctx, hash) as in the exact mode;
//     for each new position the common prefixes with its K predecessors in the run are tabulated once
//     (`rows`), so a round costs one row + a window of the member bitmap per position.
//   * afterwards the item boundaries are frozen and sources are assigned: every match takes the NEWEST
//     item start of its run whose common prefix covers its length.  With that rule the lengths referring to
//     one source ascend, so len >= len_min holds by construction (the argument of src/matcher.rs:32-50).
//     A match without such a source is cut to the longest prefix some item start offers and the rest of
//     its span is re-parsed (item starts are only ever added); WORD items are checked against the exact
//     predictor state.  Repeat until nothing changes; then the unchanged post stage encodes the items.
//
// Parity for this mode (BASELINE.json north_star): the stream decodes bit-exactly with the reference
// decoder and its size is within +-0.5 % of the reference encoder's at the same level -- not an identical
// parse.  tests/model/fast_model.c is the CPU model the design was measured with.
#pragma once
#include "orz_kernels.h"
#include "orz_parse.h"

namespace orz {

#if !defined(__HIPCC__)
inline unsigned long long g_far_stats[4] = {0, 0, 0, 0};  // (host emulation only: list scans, records read, records that went to the window, trips of the walk below the window)
#endif

constexpr uint32_t kSub = 4096;                    // positions per ordinal subtile (= path chunk)
constexpr uint32_t kSeg64 = 64;                    // positions per path segment
constexpr uint32_t kNSub = kNewMax / kSub + 2;
constexpr uint32_t kEntries = 240;                 // a path enters a chunk / tile within its first 240 positions

constexpr uint32_t kFastK = 32;                    // run predecessors tabulated per position (half a word of the member bitmap; 64: twice the
                                                   // table and its build for -0.05 % of output -- deeper candidates come from the lists)
constexpr uint32_t kHistSub = (kPre + 1) / kSub;   // unified subtiles of the history: window offset x lies in subtile (x + 1) >> 12
constexpr uint32_t kRingMargin = 4;                // item starts the repairs may still add between a source and its reference
constexpr uint32_t kFastTile = 262144;             // default Gauss-Seidel tile (positions) and rounds per tile.  Measured on a full block,
constexpr uint32_t kFastRounds = 4;                // emulator, vs the oracle: text -l1 256 K x 3 / x 4: -0.00 / -0.04 %, 512 K x 3: +0.2 %;
constexpr uint32_t kSettledTile = 393216;          // the schedule of a block that follows a block of settled, text-like statistics
constexpr uint32_t kSettledTileDeep = 262144;      // (StreamEncoder::fast_parse, round 6): three rounds cost text nothing (100 MB, -l1 / -l2:
constexpr uint32_t kSettledRounds = 3;             // 256 K x 4 -0.029 / +0.159 %, 256 K x 3 -0.028 / +0.150 %, 384 K x 3 +0.074 / +0.230 %, 512 K x 3 +0.154 /
                                                   // +0.309 %; zeros with noise -l2: +0.36 / +0.44 / +0.78 / +1.37 %); kernel time of a block 31.7 -> 31.0 /
                                                   // 29.9 / 30.1 ms.  -l2 keeps the 256 K tile (its band is the tighter one).
                                                   // zeros + noise -l2 (one hot context, item starts that depend on each other over long
                                                   // distances): 256 K x 3 / x 4 / x 5: +0.81 / +0.46 / +0.40 %, 512 K x 4: +1.5 %

struct FastArgs {
    const uint8_t* win;
    uint32_t len, n;            // window end, new bytes
    uint32_t K, depth, lazy1, lazy2, tile;
    uint32_t dmax;              // max(depth, lazy1, lazy2)
    uint32_t nent, nk;          // slots of the candidate lists / of the word-predictor lists
    // static per block
    const uint32_t *idx, *epos, *kidx, *kpos, *krun;
    const uint8_t* rows;        // [n][K] common prefix with each of the K predecessors in the run
    const uint8_t* rlen;        // [n] min(255, slots of the run below the position)
    const uint64_t* rdist;      // [n] eight distance codes: how far back the run predecessors number 1, 2, 4, 8, 16, 32, 48, 64 lie
    const uint64_t* wmask;      // [n] word predictor: which of the 64 list slots below the position hold its own next two bytes
    const uint16_t* kmeta;      // [n] list slots below in the same hash2 run (0..64) | slot below is p-1 << 7 | words[] snapshot predicts << 8
    const uint16_t* kw;         // [n+1] the two bytes at each word-list slot's position
    const uint8_t* wsnap;       // words[] at the block start
    const uint32_t* ORD;        // exact ring ordinals of history item starts
    const uint64_t* stext;      // [nent][2] slot records: the 12 leading bytes of the slot's position | the position << 96 (one 16-byte load
                                // gives a candidate's text and where it lies)
    const uint32_t* runstart;   // first slot of each (ctx, hash) run
    const uint32_t* hpre;       // [kHistSub + 1][256] history item starts per ctx before each unified subtile; [kHistSub] = all of them
    uint32_t far;               // slots searched beyond the tabulated K through the bitmap (FastSource; FastEval for item starts not in the lists yet)
    uint32_t near;              // item starts a scan takes from there (0 = none)
    uint32_t near1;             // ... in a tile's FIRST round
    uint32_t extra;             // candidates a scan may look at beyond the reference's depth (window + below the window + lists)
    // compact lists: per (ctx, hash) run the records of its FINAL item starts (history, then the tiles that had their last
    // round, appended by FastRetire) side by side from the run's first slot on, oldest first
    uint64_t* cl;               // [nent][2] records like stext
    uint32_t *ccnt, *cnew;      // [keys] records in each run's list / appended by the running FastRetire
    uint32_t rounds;            // rounds per tile
    uint32_t* farv;             // [n+8] what a position's last list scan found: len | lz1 << 8 | lz2 << 16 | ro510 << 24 | valid << 25
    // dynamic
    uint64_t *vbits, *kbits;    // item-start / word-update bitmaps in slot order
    uint64_t* v1;               // bit per non-zero word of vbits (V1Build once per parse, then kept in step by FastFlip)
    uint64_t* k1;               // the same for kbits (zeroed with it; only ever set: a stale bit costs a search one wasted load)
    uint64_t* tbits;            // [n/64+2] repair stage: positions whose item-start bit or end type was rewritten since the last flip
    uint32_t* ev;               // [n+8] best len | lz1 << 8 | lz2 << 16 | lwm << 24 | ro510 << 25
    uint8_t *ty, *nl, *pt;      // [n+264] decision type, advance, type of the item ending at the position
    uint64_t* sbits;            // [n/64+8] item starts of the current path, bit per new position
    uint8_t *mfb, *efb;         // [n+264] what vbits / kbits currently hold for each position
    uint8_t* dirty;             // [n+264] the position's candidates changed since it was last evaluated (set by FastFlip)
    uint32_t* hz;               // [kNSub][256][4] ring horizons per (subtile, ctx): oldest window offset still within 4094 / 510 item starts,
                                // and the two values one step earlier (what the evaluations of the previous step saw)
    uint8_t *x0, *x1, *x2;      // path maps: per position, per (chunk, entry), per (tile, entry)
    uint32_t *centry, *tentry;  // path entry of each chunk / tile
    uint32_t *cm, *cp;          // [kNSub][256] item starts per (subtile, ctx) and their exclusive prefix (+ carried totals)
    uint32_t* nchg;
    const uint32_t* nentp;      // -> FastCtl::nent
    uint32_t dbg;               // experiment switches (ORZ_FAST_DBG): 1 = evaluate every position every round, 64 = collect `stats`
    unsigned long long* stats;  // [32] counters / wall-clock ticks of kernel phases when dbg & 64 (diagnostics only)
};

ORZ_D uint32_t fast_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
// 64 bits of a bitmap starting at bit `start` (bits below zero read as 0)
ORZ_D uint64_t bits_at(const uint64_t* bm, int64_t start) {
    if (start <= -64) return 0;
    if (start < 0) return bm[0] << (uint32_t)(-start);
    const uint32_t sh = (uint32_t)(start & 63);
    const uint64_t lo = bm[start >> 6] >> sh;
    return sh ? lo | (bm[(start >> 6) + 1] << (64 - sh)) : lo;
}
ORZ_D void atom_xor64(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicXor((unsigned long long*)p, (unsigned long long)v);
#else
    __atomic_fetch_xor(p, v, __ATOMIC_RELAXED);
#endif
}
ORZ_D uint64_t atom_fetch_xor64(uint64_t* p, uint64_t v) {  // returns the old word
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)atomicXor((unsigned long long*)p, (unsigned long long)v);
#else
    return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED);
#endif
}
ORZ_D uint32_t atom_fetch_add32(uint32_t* p, uint32_t v) {  // returns the old value
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
#endif
}
ORZ_D void atom_sub32(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicSub(p, v);
#else
    __atomic_fetch_sub(p, v, __ATOMIC_RELAXED);
#endif
}

// Reads and writes that are MEANT to race with other threads of the same launch (the race check of tests/race reports every
// other one): on the device plain accesses, on the host relaxed atomics.  Each use says why the race is harmless.
ORZ_D uint64_t racy_load64(const uint64_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    return __atomic_load_n(p, __ATOMIC_RELAXED);
#endif
}
ORZ_D void racy_store8(uint8_t* p, uint8_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *p = v;
#else
    __atomic_store_n(p, v, __ATOMIC_RELAXED);
#endif
}
ORZ_D int popc64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll((unsigned long long)v);
#else
    return __builtin_popcountll(v);
#endif
}
ORZ_D uint32_t clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)v);
#else
    return (uint32_t)__builtin_clz(v);
#endif
}
// Position distances as 8-bit codes: exact below 16, then eight steps per octave.  A distance is coded rounded UP, a
// budget rounded DOWN, so code(distance) <= code(budget) implies distance <= budget (never the other way round by
// more than one step, 1/8 of the value).
ORZ_D uint32_t dist_code_up(uint32_t d) {
    if (d < 16) return d;
    const uint32_t e = 31 - clz32(d), sh = e - 3;
    return 16 + (e - 4) * 8 + ((d >> sh) & 7) + ((d & ((1u << sh) - 1)) ? 1 : 0);
}
ORZ_D uint32_t dist_code_down(uint32_t d) {
    if (d < 16) return d;
    const uint32_t e = 31 - clz32(d), sh = e - 3;
    return 16 + (e - 4) * 8 + ((d >> sh) & 7);
}
// the run predecessors whose distance is sampled (0 = the newest), and how many predecessors a sample vouches for
ORZ_D uint32_t dist_sample(uint32_t m) { return m < 4 ? (1u << m) - 1 : (m == 4 ? 15 : (m == 5 ? 31 : (m == 6 ? 47 : 63))); }
// Newest predecessors of a position that lie within `budget` positions, from its eight distance codes: `sure` of them
// certainly do, the ones from `limit` on certainly do not; what lies between (at most two sample intervals, only when
// the budget ends inside the window) is for the caller to settle with the positions themselves.
struct DistBracket { uint32_t sure, limit; };
ORZ_D DistBracket dist_valid(uint64_t codes, uint32_t budget) {
    const uint32_t bd = dist_code_down(budget), bu = dist_code_up(budget);
    DistBracket b{0, 64};
    bool open = true;
#pragma unroll
    for (uint32_t m = 0; m < 8; m++) {
        const uint32_t code = (uint32_t)((codes >> (8 * m)) & 0xff);
        if (code <= bd) b.sure = dist_sample(m) + 1;
        else if (code > bu && open) { b.limit = dist_sample(m); open = false; }
    }
    return b;
}

// Several buffers zeroed by ONE launch (round 6): the resets in front of a block's rounds were 17 fill dispatches, each a launch
// that under eight encoders waits ~118 us for its turn whatever its size (rocprofv3, round 5) -- and 244 fill dispatches a block
// in all.  Thread per 16-byte unit over the ranges laid end to end; a range may start and end anywhere (its first and last unit
// are written byte by byte where they reach outside it).
struct ZeroRanges {
    static constexpr int kMax = 24;
    uint8_t* base[kMax];      // start of each range rounded down to 16
    uint32_t lead[kMax];      // bytes of the first unit in front of the range
    uint64_t bytes[kMax];     // length of the range
    uint32_t ustart[kMax + 1];  // first unit of each range
    int n = 0;
    void add(void* p, size_t len) {  // (host side, while the launch is put together)
        if (!len) return;
        if (n >= kMax) return;  // (callers check full() and launch in two goes)
        const uintptr_t a = reinterpret_cast<uintptr_t>(p), a0 = a & ~(uintptr_t)15;
        base[n] = reinterpret_cast<uint8_t*>(a0);
        lead[n] = (uint32_t)(a - a0);
        bytes[n] = len;
        if (n == 0) ustart[0] = 0;
        ustart[n + 1] = ustart[n] + (uint32_t)((lead[n] + len + 15) / 16);
        n++;
    }
    bool full() const { return n >= kMax; }
    size_t units() const { return n ? ustart[n] : 0; }
    ORZ_HD void operator()(size_t tid) const {
        if (!n || tid >= ustart[n]) return;
        int r = 0;
#pragma unroll 1
        for (int k = 1; k < n; k++) r += tid >= ustart[k] ? 1 : 0;
        const uint64_t u = tid - ustart[r], lo = u * 16, first = lead[r], end = first + bytes[r];
        uint8_t* q = base[r] + lo;
        if (lo >= first && lo + 16 <= end) {
            uint64_t* w = reinterpret_cast<uint64_t*>(q);
            w[0] = 0; w[1] = 0;
        } else {
            for (uint32_t b = 0; b < 16; b++)
                if (lo + b >= first && lo + b < end) q[b] = 0;
        }
    }
};

// runs (ctx, hash) that gained an item start in a repair pass: one bit per run key
ORZ_D void mark_run(const uint8_t* win, uint64_t* rdirty, uint32_t x) {
    const uint32_t key = bucket_key(win, x);
    atom_or64(&rdirty[key >> 6], 1ull << (key & 63));
}

// ---- static per block --------------------------------------------------------------------------------
// One wavefront per 64 slots = one word of the item-start bitmap: history slots are item starts for good (the word is written
// whole from a ballot: no fill of the bitmap before, no atomics -- as a thread per slot this was 64 atomic ORs on one word per
// wavefront); run depth of each new position.
struct FastSlotInitWave {
    const uint32_t *epos, *keys, *runstart;
    uint32_t nent;
    uint64_t* vbits;
    uint8_t* rlen;
    // ... and, in the same pass over the slots (round 5: one kernel less, the slot positions and keys read once), the slot records
    // (12 text bytes + the position, the heads of the compact lists, their counters); win == nullptr: none
    const uint8_t* win = nullptr;
    uint64_t *stext = nullptr, *cl = nullptr;
    uint32_t* ccnt = nullptr;
    static size_t lds_bytes() { return 0; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        const uint32_t j = w.block() * 64 + w.lane();
        const bool valid = j < nent;
        const uint32_t p = valid ? epos[j] : ~0u;
        const uint64_t m = w.ballot(valid && p < kPre);
        if (w.lane() == 0) vbits[w.block()] = m;
        if (!valid || !keys) return;  // (keys == nullptr: bitmap only, the run depths are already there)
        const uint32_t key = keys[j], rs = runstart[key];
        if (win) {
            const uint64_t lo = ldu64(win + p), hi = (uint64_t)ldu32(win + p + 8) | ((uint64_t)p << 32);
            stext[2 * (size_t)j] = lo; stext[2 * (size_t)j + 1] = hi;
            if (p < kPre) {
                cl[2 * (size_t)j] = lo; cl[2 * (size_t)j + 1] = hi;
                // the history slots lead their run (stable sort: history positions first): the last of them knows how many there are
                if (j + 1 == nent || keys[j + 1] != key || epos[j + 1] >= kPre) ccnt[key] = j + 1 - rs;
            }
        }
        if (p < kPre) return;
        const uint32_t d = j - rs;
        rlen[p - kPre] = (uint8_t)(d < 255 ? d : 255);
    }
};
struct FastKw {
    const uint8_t* win;
    const uint32_t* kpos;
    uint32_t nk;
    uint16_t* kw;
    ORZ_HD void operator()(size_t s) const {
        if (s >= nk) return;
        const uint32_t u = kpos[s];
        kw[s] = (uint16_t)(win[u] | (win[u + 1] << 8));
    }
};
// Static half of the word predictor (src/lz.rs:132-133): the prediction for p is the two bytes behind the newest
// words[] update among the earlier positions of its hash2 run.  Which positions update is decided by the parse (kbits);
// whether a given one would predict p's bytes is not -- tabulated here once per block for the 64 list slots below p, so
// that a round needs the bitmap window only.
struct FastWordMasks {  // one wavefront per 64 word-list slots, the two bytes of the 64 + 64 slots involved staged in LDS
    const uint32_t *kpos, *kkeys, *krun;
    const uint16_t* kw;
    const uint8_t* wsnap;
    uint32_t nk;
    uint64_t* wmask;
    uint16_t* kmeta;
    static size_t lds_bytes() { return 128 * 2; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        uint16_t* kwL = (uint16_t*)w.lds();
        const uint32_t lane = w.lane();
        const int64_t base = (int64_t)w.block() * 64 - 64;  // slot of LDS entry 0
        for (uint32_t e = lane; e < 128; e += 64) {
            const int64_t s = base + e;
            kwL[e] = s >= 0 && s < (int64_t)nk ? kw[s] : 0;
        }
        w.sync();
        const uint32_t s = w.block() * 64 + lane;
        if (s >= nk) return;
        const uint32_t u = kpos[s];
        if (u < kPre) return;  // (the list starts one position before the block)
        const uint32_t key = kkeys[s], rk = fast_min(64u, s - krun[key]);
        const uint32_t me = 64 + lane, wv = kwL[me];
        uint64_t m = 0;
        for (uint32_t t = 0; t < rk; t++) m |= (uint64_t)(kwL[me - 1 - t] == wv) << (63 - t);
        const uint32_t excl = rk && kpos[s - 1] == u - 1;  // the slot right below is u = p-1: its update comes too late
        const uint32_t snap = ((uint32_t)wsnap[key * 2] | ((uint32_t)wsnap[key * 2 + 1] << 8)) == wv;
        wmask[u - kPre] = m;
        kmeta[u - kPre] = (uint16_t)(rk | (excl << 7) | (snap << 8));
    }
};
// History item starts per (unified subtile, ctx): one wavefront per subtile, counters in LDS (the ring horizons of a
// round reach back into the history: FastHorizon)
struct HistCountWave {
    const uint8_t* win;
    const uint8_t* S;
    uint32_t* hcm;  // [kHistSub][256]
    static size_t lds_bytes() { return 256 * 4; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        uint32_t* cnt = (uint32_t*)w.lds();
        const uint32_t u = w.block(), lane = w.lane();
        for (uint32_t c = lane; c < 256; c += 64) cnt[c] = 0;
        w.sync();
        const uint32_t x0 = u * kSub + lane * 64;  // positions x0 - 1 .. x0 + 62
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t x = x0 + k - 1;
            if (x0 + k == 0 || x < 1 || x >= kPre) continue;  // position 0 is dead (src/matcher.rs:85)
            if (S[x]) atom_add32(&cnt[hash1(win, x - 1)], 1);
        }
        w.sync();
        for (uint32_t c = lane; c < 256; c += 64) hcm[(size_t)u * 256 + c] = cnt[c];
    }
};
// Exclusive prefix down the columns of a [rows][256] table in three small launches (groups of 64 rows): group sums,
// their prefix per column, then the rows of each group.  out[0][c] must hold the base; out gets rows + 1 rows.
struct FastCtl;
ORZ_HD bool passes_done(const FastCtl* ctl);  // (defined with FastCtl below: a repair-stage kernel returns at once when the passes are done)
struct ColScanGroups {
    const uint32_t* in;
    uint32_t rows;
    uint32_t* gsum;  // [groups][256]
    const FastCtl* ctl = nullptr;
    ORZ_HD void operator()(size_t tid) const {
        if (passes_done(ctl)) return;
        const uint32_t c = (uint32_t)(tid & 255), g = (uint32_t)(tid >> 8);
        if (g * 64 >= rows) return;
        uint32_t v = 0;
        for (uint32_t r = g * 64; r < rows && r < g * 64 + 64; r++) v += in[(size_t)r * 256 + c];
        gsum[(size_t)g * 256 + c] = v;
    }
};
struct ColScanTop {
    uint32_t* gsum;
    uint32_t rows;
    const uint32_t* out;  // out[0][c] = base
    const FastCtl* ctl = nullptr;
    ORZ_HD void operator()(size_t c) const {
        if (c >= 256 || passes_done(ctl)) return;
        uint32_t v = out[c];
        for (uint32_t g = 0; g * 64 < rows; g++) {
            const uint32_t t = gsum[(size_t)g * 256 + c];
            gsum[(size_t)g * 256 + c] = v;
            v += t;
        }
    }
};
struct ColScanRows {
    const uint32_t *in, *gsum;
    uint32_t rows;
    uint32_t* out;
    const FastCtl* ctl = nullptr;
    ORZ_HD void operator()(size_t tid) const {
        if (passes_done(ctl)) return;
        const uint32_t c = (uint32_t)(tid & 255), g = (uint32_t)(tid >> 8);
        if (g * 64 >= rows) return;
        uint32_t v = gsum[(size_t)g * 256 + c];
        for (uint32_t r = g * 64; r < rows && r < g * 64 + 64; r++) {
            out[(size_t)r * 256 + c] = v;
            v += in[(size_t)r * 256 + c];
        }
        if (g * 64 + 64 >= rows) out[(size_t)rows * 256 + c] = v;
    }
};
// Slot record = 12 text bytes + the position.  rec_lcp: common prefix of two records' texts, 12 = all twelve agree
// (the rest comes from the window).
constexpr uint32_t kRecText = 12;
ORZ_D uint32_t rec_pos(uint64_t hi) { return (uint32_t)(hi >> 32); }
ORZ_D uint32_t rec_lcp(uint64_t lo_a, uint64_t hi_a, uint64_t lo_b, uint64_t hi_b) {
    const uint64_t x0 = lo_a ^ lo_b;
    if (x0) return (uint32_t)ctz64(x0) >> 3;
    const uint32_t x1 = (uint32_t)(hi_a ^ hi_b);
    return x1 ? 8 + ((uint32_t)ctz64((uint64_t)x1) >> 3) : kRecText;
}
// Common prefixes of a new position with its K predecessors in the (ctx, hash) run.
// One wavefront per 64 consecutive slots: the records (12 leading bytes + position) of the 64 + K slots involved are
// staged in LDS once, so a pair costs one LDS read; only pairs that agree on all 12 bytes go to the window.
// The rows leave through LDS as well, 64 columns at a time, so that every 64-byte piece of a row is written by
// four neighbouring lanes in one go (a lane storing its own row eight bytes at a time costs a partial line per store).
struct FastRowsWave {
    const uint8_t* win;
    const uint32_t* epos;
    const uint64_t* stext;
    const uint8_t* rlen;
    uint32_t nent, K;  // K = 32 or a multiple of 64
    uint8_t* rows;
    uint64_t* rdist;   // [n] distance codes of the sampled predecessors (dist_valid)
    static constexpr uint32_t kOutStride = 72;  // bytes per lane in the staging tile (64 + pad against bank conflicts)
    static size_t lds_bytes(uint32_t K) { return (size_t)(64 + K) * 16 + 64 * kOutStride; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        uint64_t* t0 = (uint64_t*)w.lds();            // [64+K] bytes 0..7
        uint64_t* t1 = t0 + (64 + K);                 // [64+K] bytes 8..11 | position << 32
        uint8_t* outL = (uint8_t*)(t1 + (64 + K));    // [64][kOutStride]
        const uint32_t lane = w.lane();
        const int64_t base = (int64_t)w.block() * 64 - K;  // slot of LDS entry 0
        for (uint32_t e = lane; e < 64 + K; e += 64) {
            const int64_t s = base + e;
            uint64_t a = 0, b = 0;
            if (s >= 0 && s < (int64_t)nent) { a = stext[2 * s]; b = stext[2 * s + 1]; }
            t0[e] = a; t1[e] = b;
        }
        w.sync();
        const uint32_t me = K + lane;
        const uint32_t p = rec_pos(t1[me]);
        const bool mine = (int64_t)w.block() * 64 + lane < (int64_t)nent && p >= kPre;
        const uint32_t r = mine ? fast_min(K, rlen[p - kPre]) : 0;
        const uint64_t a0 = t0[me], a1 = t1[me];
        const uint64_t a2 = mine ? ldu64(win + p + kRecText) : 0;  // the position's bytes 12..19
        if (mine) {  // how far back the sampled predecessors lie (a sample beyond the run's depth stands for its oldest member)
            uint64_t codes = 0;
            for (uint32_t m = 0; m < 8; m++) {
                const uint32_t k = r ? fast_min(dist_sample(m), r - 1) : 0;
                const uint32_t code = r ? dist_code_up(p - rec_pos(t1[me - 1 - k])) : 255u;
                codes |= (uint64_t)code << (8 * m);
            }
            rdist[p - kPre] = codes;
        }
        const uint32_t cols = K < 64 ? K : 64;       // columns per pass (K = 32, or a multiple of 64)
        const uint32_t parts = cols / 16;            // lanes that share one row's piece of a pass
        for (uint32_t c0 = 0; c0 < K; c0 += cols) {
            for (uint32_t k0 = 0; k0 < cols; k0 += 8) {
                // eight pairs at a time: the prefixes inside the records first (LDS only), then -- for the pairs that agree on all
                // twelve text bytes -- the candidates' next eight bytes fetched TOGETHER; only pairs that agree on twenty bytes walk
                // the window one after the other (before: every pair beyond twelve was a chain of its own, 32 pairs a lane in a row)
                uint32_t ll[8], lq[8];
                uint64_t yy[8];
#pragma unroll
                for (uint32_t kk = 0; kk < 8; kk++) {
                    const uint32_t k = c0 + k0 + kk;
                    ll[kk] = 0; lq[kk] = 0;
                    if (k < r) {
                        const uint32_t e = me - 1 - k;
                        const uint64_t hi = t1[e];
                        ll[kk] = rec_lcp(t0[e], hi, a0, a1);
                        lq[kk] = rec_pos(hi);
                    }
                }
#pragma unroll
                for (uint32_t kk = 0; kk < 8; kk++) yy[kk] = ll[kk] == kRecText ? ldu64(win + lq[kk] + kRecText) : 0;
                uint64_t pack = 0;
#pragma unroll
                for (uint32_t kk = 0; kk < 8; kk++) {
                    uint32_t l = ll[kk];
                    if (l == kRecText) {
                        const uint64_t d = yy[kk] ^ a2;
                        l = d ? kRecText + ((uint32_t)ctz64(d) >> 3)
                              : kRecText + 8 + lcp240u(win + lq[kk] + kRecText + 8, win + p + kRecText + 8, kMaxLen - kRecText - 8);
                    }
                    pack |= (uint64_t)l << (8 * kk);
                }
                *reinterpret_cast<uint64_t*>(outL + lane * kOutStride + k0) = pack;
            }
            w.sync();
            for (uint32_t it = 0; it < parts; it++) {  // 64 / parts rows per pass, `parts` lanes per row piece of 16 * parts bytes
                const uint32_t row = it * (64 / parts) + lane / parts, part = lane % parts;
                const uint32_t pr = rec_pos(t1[K + row]);
                if ((int64_t)w.block() * 64 + row < (int64_t)nent && pr >= kPre) {  // (whole row pieces: parts of a line cost a read-modify-write)
                    const uint64_t v0 = *reinterpret_cast<const uint64_t*>(outL + row * kOutStride + part * 16);
                    const uint64_t v1 = *reinterpret_cast<const uint64_t*>(outL + row * kOutStride + part * 16 + 8);
                    uint64_t* dst = reinterpret_cast<uint64_t*>(rows + (size_t)(pr - kPre) * K + c0 + part * 16);
                    dst[0] = v0; dst[1] = v1;
                }
            }
            w.sync();
        }
    }
};


// common prefix of position p (its record in a0/a1) with the position of slot s (returned in *q): from the slot records,
// through the window only when all 12 text bytes agree
ORZ_D uint32_t far_lcp(const FastArgs& a, uint32_t p, uint64_t a0, uint64_t a1, uint32_t s, uint32_t* q) {
    const uint64_t lo = a.stext[2 * (size_t)s], hi = a.stext[2 * (size_t)s + 1];
    *q = rec_pos(hi);
    const uint32_t l = rec_lcp(lo, hi, a0, a1);
    return l < kRecText ? l : kRecText + lcp240u(a.win + *q + kRecText, a.win + p + kRecText, kMaxLen - kRecText);
}

// Item starts among the slots [lo, top), newest first, through the summary level: visit(slot) returns false to stop.
template <class V>
ORZ_D void far_walk(const FastArgs& a, uint32_t lo, uint32_t top, V visit) {
    if (top <= lo) return;
    const uint32_t w_hi = (top - 1) >> 6, w_lo = lo >> 6;  // words of vbits that intersect the range
    for (int64_t g = (int64_t)(w_hi >> 6); g >= (int64_t)(w_lo >> 6); g--) {
        uint64_t sm = a.v1[g];
        if ((uint32_t)g == (w_hi >> 6) && (w_hi & 63) != 63) sm &= (2ull << (w_hi & 63)) - 1;
        if ((uint32_t)g == (w_lo >> 6)) sm &= ~0ull << (w_lo & 63);
        while (sm) {
            const uint32_t wb = 63 - (uint32_t)clz64(sm);
            sm &= ~(1ull << wb);
            const uint32_t wi = (uint32_t)g * 64 + wb;
            uint64_t m = a.vbits[wi];
            if (wi == w_hi && (top & 63)) m &= (1ull << (top & 63)) - 1;
            if (wi == w_lo) m &= ~0ull << (lo & 63);
            while (m) {
                const uint32_t t = 63 - (uint32_t)clz64(m);
                m &= ~(1ull << t);
                if (!visit(wi * 64 + t)) return;
            }
        }
    }
}
struct V1Build {  // thread per summary word: 64 words of the bitmap
    const uint64_t* vbits;
    uint32_t nwords;  // words of vbits
    uint64_t* v1;
    ORZ_HD void operator()(size_t g) const {
        if (g * 64 >= nwords) return;
        uint64_t m = 0;
        for (uint32_t k = 0; k < 64 && g * 64 + k < nwords; k++) m |= (uint64_t)(vbits[g * 64 + k] != 0) << k;
        v1[g] = m;
    }
};

// Common prefix of x[0..) and y[0..), capped at `cap`, 32 bytes per trip with the eight loads of a trip in flight: a match
// of a few dozen bytes costs one or two memory round trips instead of one per eight bytes.
ORZ_D uint32_t lcp_trips(const uint8_t* x, const uint8_t* y, uint32_t cap) {
    for (uint32_t off = 0; off < cap; off += 32) {
        uint64_t d[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) d[k] = ldu64(x + off + 8 * k) ^ ldu64(y + off + 8 * k);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (d[k]) { const uint32_t l = off + 8 * k + ((uint32_t)ctz64(d[k]) >> 3); return l < cap ? l : cap; }
    }
    return cap;
}
// The newest item starts among the slots [lo, top) -- at most four, from the at most four newest non-empty bitmap words of
// the two summary words that reach down from `top` (8192 slots at most): two independent summary loads, then four
// independent bitmap loads -- a chain of two dependent rounds whatever the run looks like.  Returns how many were found;
// slots[] newest first.
ORZ_D uint32_t near_members(const FastArgs& a, uint32_t lo, uint32_t top, uint32_t want, uint32_t* slots) {
    if (top <= lo || !want) return 0;
    const uint32_t w_hi = (top - 1) >> 6, w_lo = lo >> 6, g_hi = w_hi >> 6, g_lo = w_lo >> 6;
    uint64_t sm[2];
    sm[0] = a.v1[g_hi];
    sm[1] = g_hi > g_lo ? a.v1[g_hi - 1] : 0;
    if ((w_hi & 63) != 63) sm[0] &= (2ull << (w_hi & 63)) - 1;
    if (g_hi == g_lo) sm[0] &= ~0ull << (w_lo & 63);
    else if (g_hi - 1 == g_lo) sm[1] &= ~0ull << (w_lo & 63);
    uint32_t wi[4], nw = 0;
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) {
        uint64_t m = sm[h];
        while (m && nw < 4) {
            const uint32_t b = 63 - (uint32_t)clz64(m);
            m &= ~(1ull << b);
            wi[nw++] = (g_hi - h) * 64 + b;
        }
    }
    uint64_t wv[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) wv[k] = k < nw ? a.vbits[wi[k]] : 0;
    uint32_t n = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k >= nw) break;
        uint64_t m = wv[k];
        if (wi[k] == w_hi && (top & 63)) m &= (1ull << (top & 63)) - 1;
        if (wi[k] == w_lo) m &= ~0ull << (lo & 63);
        while (m && n < want) {
            const uint32_t b = 63 - (uint32_t)clz64(m);
            m &= ~(1ull << b);
            slots[n++] = wi[k] * 64 + b;
        }
    }
    return n;
}

// Common prefix of position p with a candidate at q whose slot record agreed on all twelve text bytes; `a2` = p's bytes 12..19,
// `y` = the candidate's, fetched TOGETHER with those of the other candidates of the trip: most such pairs part within these
// eight bytes, and a scan was a chain of one memory round trip per candidate that got this far (a launch of FastEval lasts as
// long as its slowest scans: as a kernel of their own, densely packed, they took 98 us).  = 12 + lcp_trips(q + 12, p + 12, 228).
ORZ_D uint32_t lcp_after12(const uint8_t* win, uint32_t q, uint32_t p, uint64_t a2, uint64_t y) {
    const uint64_t d = y ^ a2;
    if (d) return kRecText + ((uint32_t)ctz64(d) >> 3);
    return kRecText + 8 + lcp_trips(win + q + kRecText + 8, win + p + kRecText + 8, kMaxLen - kRecText - 8);
}

// ---- one round: the positions of the active range decide from the snapshot ----------------------------
// A position is evaluated in its tile's first two rounds, and afterwards only when the item starts it looks at changed
// (FastFlip marks it dirty) or when its tile's scan is due.  Its candidates, newest first (find_match walks the hash
// chain of the item starts of (ctx, hash) newest first, src/matcher.rs:135-192), come from three places:
//   1. the K run predecessors tabulated in its row: the window of the item-start bitmap (slot order) against the row --
//      static data read in position order and one window of a bitmap, no load depends on another one except bitmap <- idx;
//   and, when the run is deeper than K (a scan: in the tile's first round and again in its last one, when every earlier
//   tile is final; the answer is remembered in between),
//   2. the next few item starts below the window that are not in the lists yet (tiles still in their rounds), found
//      through the bitmap's summary level -- a short chain of dependent loads, which is what sparse deep runs (zero
//      runs, long periods: an item start every few hundred slots) live on;
//   3. the run's compact list -- the records of its FINAL item starts (history + tiles that had their last round) side
//      by side, so the newest of them are one contiguous read whose address follows from the run's counter.  The records
//      of item starts that the window shows as well (set bits of slots below `cline`) are skipped.
// All parts together may look at more than `depth` candidates -- the depth is the reference's speed limit, not a rule of
// the format -- which is why this parse can come out smaller than the reference's.
#if !defined(__HIPCC__)
inline unsigned long long g_eval_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (host emulation only: positions visited, evaluated, by round-1 / dirty / scan-due)
// (host emulation only, ORZ_SCAN_HIST) scans by the candidates the window had counted: [0] first-round scans, [1] last-round ones,
// [2] last-round ones whose answer differs from the remembered one, [3] the same by item starts in the window
inline unsigned long long g_scan_hist[4][64] = {};
#endif
struct FastEval {
    FastArgs a;
    uint32_t lo, hi;    // window offsets [lo, hi)
    uint32_t r1lo;      // positions >= r1lo have not been evaluated in this parse yet (the tile in its first round, and beyond)
    uint32_t r2lo;      // positions >= r2lo are evaluated whatever their flags say (the tiles in their first two rounds: after a
                        // tile's first round nearly every position with a predecessor inside the tile is dirty anyway)
    uint32_t step;      // tile t runs its round step - t
    uint32_t cline;     // window offset below which every item start is in the compact lists (the tiles retired so far)
    // newest run predecessors (of the first r) at or after `horizon`, from the distance bracket; an uncertain stretch is
    // settled with the positions themselves (independent loads; validity is monotone, so counting is enough)
    ORZ_D uint32_t count_from(DistBracket d, uint32_t r, uint32_t j, uint32_t horizon) const {
        const uint32_t kend = fast_min(r, d.limit);
        uint32_t v = d.sure;
        for (uint32_t k0 = d.sure; k0 < kend; k0 += 8) {
            uint32_t q[8];
#pragma unroll
            for (uint32_t b = 0; b < 8; b++) q[b] = k0 + b < kend ? a.epos[j - 1 - (k0 + b)] : 0;
#pragma unroll
            for (uint32_t b = 0; b < 8; b++) v += k0 + b < kend && q[b] >= horizon;
        }
        return v;
    }
    // The candidates beyond the tabulated window for position p (parts 2 and 3 of the kernel's comment): what the scan found, as
    // farv keeps it.  `seen` = candidates the window has counted already.
    ORZ_D uint32_t scan_far(uint32_t p, uint32_t i, uint32_t c, uint32_t h4, uint32_t h5, uint64_t codes, uint32_t j, uint32_t r, uint64_t wbits,
                            uint32_t seen, uint32_t rnd, bool first) const {
        const uint8_t* win = a.win;
        (void)i; (void)first;
        uint32_t fv;
        {
                const uint64_t a0 = ldu64(win + p), a1 = (uint64_t)ldu32(win + p + 8), a2 = ldu64(win + p + kRecText);
                uint32_t fbest = 0, f510 = 0, fm1 = 0, fm2 = 0, s = seen;
                bool fin = false;
                // one older candidate by the serial rules of find_match; false = nothing older matters
                auto take = [&](uint32_t q, uint32_t l) -> bool {
                    if (l > fbest || (s < a.lazy1 && l > fm1) || (s < a.lazy2 && l > fm2)) {
                        if (q < h4) return false;  // left the ring (asked only for candidates that matter): so did everything older
                        if (l > fbest) { fbest = l; f510 = q >= h5; }
                        if (s < a.lazy1 && l > fm1) fm1 = l;
                        if (s < a.lazy2 && l > fm2) fm2 = l;
                    }
                    s++;
                    return l != kMaxLen;
                };
                const uint32_t key = c * kHash + hash_entry(win + p);
                // (the records a retiring tile appended in the step before are counted in cnew until the next FastDecide's grid
                // folds them into ccnt -- PathUpWave's piggy-backed FastRetireDone)
                const uint32_t rs = a.runstart[key], cnt = a.ccnt[key] + a.cnew[key];
                // ---- 2. item starts below the window that are not in the lists yet: four at a time -- their slots from the
                // bitmap, their records side by side, then in order -- until the lists take over (below `cline`), the budget
                // is spent or `near` of them were looked at.  (One hot context -- zeros with noise -- has its whole ring
                // inside the tiles that are still in their rounds: the lists' records lie outside it and this walk is all.)
                const uint32_t nearn = rnd <= 1 ? a.near1 : a.near;
                if (nearn && p > cline) {
                    uint32_t top = j - kFastK;
                    const uint32_t lo2 = top - rs > a.far ? top - a.far : rs;
                    uint32_t left = fast_min(nearn, a.depth + a.extra > s ? a.depth + a.extra - s : 0);
                    bool more = left != 0;
                    while (more && !fin) {
                        uint32_t sl[4];
                        const uint32_t ns = near_members(a, lo2, top, fast_min(left, 4u), sl);
                        uint64_t x0[4], x1[4];
#pragma unroll
                        for (uint32_t b = 0; b < 4; b++) {
                            const uint32_t s2 = b < ns ? sl[b] : j;
                            x0[b] = a.stext[2 * (size_t)s2];
                            x1[b] = a.stext[2 * (size_t)s2 + 1];
                        }
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
                        g_far_stats[3]++;
#endif
                        uint32_t ql[4], ll[4];
                        uint64_t yy[4];
#pragma unroll
                        for (uint32_t b = 0; b < 4; b++) { ql[b] = rec_pos(x1[b]); ll[b] = rec_lcp(x0[b], x1[b], a0, a1); }
#pragma unroll
                        for (uint32_t b = 0; b < 4; b++) yy[b] = b < ns && ll[b] == kRecText ? ldu64(win + ql[b] + kRecText) : 0;
#pragma unroll
                        for (uint32_t b = 0; b < 4; b++) {
                            if (b >= ns || fin || !more) break;
                            const uint32_t q = ql[b];
                            if (q < cline) { more = false; break; }  // from here on the lists have them
                            uint32_t l = ll[b];
                            if (l == kRecText) l = lcp_after12(win, q, p, a2, yy[b]);
                            if (!take(q, l)) fin = true;
                            left--;
                        }
                        if (ns < 4 || left == 0) more = false;
                        else top = sl[3];  // go on below the fourth
                    }
                }
                // ---- 3. the run's compact list, newest record first; the window's own item starts below the line lead it
                if (!fin) {
                    const uint32_t nabove = fast_min(r, dist_valid(codes, p > cline ? p - cline : 0).limit);  // tabulated predecessors at or after the line (never too few)
                    const uint32_t skip = nabove < 64 ? (uint32_t)popc64(wbits & (~0ull >> nabove)) : 0;
                    const uint32_t avail = cnt > skip ? cnt - skip : 0;
                    const uint64_t* top = a.cl + 2 * ((size_t)rs + avail);
                    const uint32_t room = a.depth + a.extra > s ? a.depth + a.extra - s : 0;
                    const uint32_t want = fast_min(room, avail);
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
                    g_far_stats[0]++;
#endif
                    constexpr uint32_t kB = 8;  // records per trip, all loads of a trip in flight
                    for (uint32_t k0 = 0; k0 < want && !fin; k0 += kB) {
                        uint64_t x0[kB], x1[kB];
#pragma unroll
                        for (uint32_t b = 0; b < kB; b++) {
                            const uint32_t k = k0 + b < want ? k0 + b : k0;
                            x0[b] = top[-2 * (int64_t)(k + 1)];
                            x1[b] = top[-2 * (int64_t)(k + 1) + 1];
                        }
                        uint32_t ql[kB], ll[kB];
                        uint64_t yy[kB];
#pragma unroll
                        for (uint32_t b = 0; b < kB; b++) { ql[b] = rec_pos(x1[b]); ll[b] = rec_lcp(x0[b], x1[b], a0, a1); }
#pragma unroll
                        for (uint32_t b = 0; b < kB; b++) yy[b] = k0 + b < want && ll[b] == kRecText ? ldu64(win + ql[b] + kRecText) : 0;
#pragma unroll
                        for (uint32_t b = 0; b < kB; b++) {
                            if (k0 + b >= want || fin) break;
                            const uint32_t q = ql[b];
                            uint32_t l = ll[b];
                            if (l == kRecText) l = lcp_after12(win, q, p, a2, yy[b]);
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
                            g_far_stats[1]++;
                            if (l >= kRecText) g_far_stats[2]++;
#endif
                            if (!take(q, l)) fin = true;
                        }
                    }
                }
                fv = fbest | (fm1 << 8) | (fm2 << 16) | (f510 << 24) | (1u << 25);
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
                {
                    const uint32_t sb = seen < 63 ? seen : 63;
                    if (first) g_scan_hist[0][sb]++;
                    else if (rnd == a.rounds) {
                        g_scan_hist[1][sb]++;
                        if (a.farv[i] != fv) { g_scan_hist[2][sb]++; g_scan_hist[3][popc64(wbits) < 63 ? popc64(wbits) : 63]++; }
                    }
                }
#endif
        }
        return fv;
    }
    // Which position a thread takes.  A launch covers the active tiles oldest first, and its wavefronts are dispatched in thread
    // order: the tile in its FIRST round -- every deep position scans -- came last, so the launch ended on its heaviest wavefronts
    // with the GPU half empty.  Round 6: the newest tile's threads come first, then the oldest tile's (its last round: the second
    // scan), then the tiles between (window evaluations only), each tile's positions in order as before (coalescing is per
    // tile).  The threads are independent: same answers, another order.  (a.dbg & 2048: the old order, for measurements.)
    ORZ_D uint32_t position_of(size_t tid) const {
        const uint32_t span = hi - lo, T = a.tile;
        if ((a.dbg & 2048) || span <= T) return lo + (uint32_t)tid;
        const uint32_t nch = (span + T - 1) / T, s_last = span - (nch - 1) * T;  // chunks of T positions; the last one is short
        if (tid < s_last) return lo + (nch - 1) * T + (uint32_t)tid;
        const uint32_t u = (uint32_t)tid - s_last, k = u / T, off = u - k * T;
        const uint32_t c = k == 0 ? nch - 2 : k - 1;  // the newest full chunk, then chunks 0, 1, ... in order
        return lo + c * T + off;
    }
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t p = position_of(tid);
        if (p >= hi) return;
        const uint8_t* win = a.win;
        const uint32_t i = p - kPre;
        const uint32_t rl = a.rlen[i];
        const bool first = p >= r1lo;
        const bool longrun = rl > kFastK;
        // the round of the position's tile (0: the two positions beyond the range, looked at by the lazy rules)
        const uint32_t t = i / a.tile, rnd = step > t ? step - t : 0;
        bool scan = longrun && (rnd <= 1 || rnd == a.rounds);
        if (a.dbg & (12 | 128 | 256)) {  // (experiments: 4 = no second scan, 8 = no scans at all, 16 below = the second scan only where the path looks,
            if ((a.dbg & 8) || ((a.dbg & 4) && rnd > 1)) scan = false;  // 128 = no first-round scan, 256 = the first scan in round 2 instead of 1)
            if ((a.dbg & 128) && rnd <= 1) scan = false;
            if (a.dbg & 256) scan = longrun && (rnd == 2 || rnd == a.rounds);
        }
        if ((a.dbg & 16) && scan && rnd > 1) {
            // relevant to the current path: an item start, or one of the two positions behind an item start that found a match
            const uint64_t sw = bits_at(a.sbits, (int64_t)i - 2);  // bits i-2, i-1, i at 0, 1, 2
            const bool rel = ((sw >> 2) & 1) || (((sw >> 1) & 1) && i >= 1 && a.ty[i - 1] == kTyMatch) || ((sw & 1) && i >= 2 && a.ty[i - 2] == kTyMatch);
            if (!rel) scan = false;
        }
        const bool dirty = a.dirty[i] != 0 || (a.dbg & 1);
        const uint32_t c = hash1(win, p - 1);
        const uint32_t* hz = a.hz + ((size_t)(i / kSub) * 256 + c) * 4;
        const uint32_t h4 = hz[0], h4was = hz[2];
        const bool need = p >= r2lo || dirty || scan;
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
        g_eval_stats[0]++;
        if (first) g_eval_stats[2]++; else if (dirty) g_eval_stats[3]++; else if (scan) g_eval_stats[4]++;
#endif
        if (!need && h4was == h4) return;
        const uint64_t codes = a.rdist[i];
        const DistBracket d4 = dist_valid(codes, p > h4 ? p - h4 : 0);
        // the ring horizon of (subtile, ctx) moved since the previous step: re-evaluate where that changes which of the
        // window's predecessors count (every active position looks every step, so one step of history is enough)
        bool moved = false;
        if (!first && h4was != h4) {
            const DistBracket o4 = dist_valid(codes, p > h4was ? p - h4was : 0);
            moved = o4.sure != d4.sure || o4.limit != d4.limit || d4.sure < fast_min(fast_min(kFastK, rl), d4.limit);
        }
        if (!need && !moved) return;
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
        g_eval_stats[1]++;
#endif
        // ---- everything the evaluation reads, asked for up front: position-ordered statics, then the two bitmap windows
        const uint32_t j = a.idx[p], kj = a.kidx[p], r = fast_min(kFastK, rl);
        const uint32_t km = a.kmeta[i], rk = km & 0x7f;
        const uint64_t wm = a.wmask[i];
        const uint32_t h5 = hz[1];
        uint32_t rw[kFastK / 4];  // the row of common prefixes, in registers (only the 16-byte pieces the run depth reaches)
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(a.rows + (size_t)i * kFastK);
#pragma unroll
            for (uint32_t q = 0; q < kFastK / 16; q++) {
                if (r > q * 16) {
#pragma unroll
                    for (uint32_t d = 0; d < 4; d++) rw[q * 4 + d] = src[q * 4 + d];
                } else {
#pragma unroll
                    for (uint32_t d = 0; d < 4; d++) rw[q * 4 + d] = 0;
                }
            }
        }
        uint64_t mask = r ? bits_at(a.vbits, (int64_t)j - 64) : 0;
        uint64_t kmask = rk ? bits_at(a.kbits, (int64_t)kj - 64) : 0;
        if (r < 64) mask = r ? mask & (~0ull << (64 - r)) : 0;
        const uint64_t wbits = mask;  // the item starts among the tabulated predecessors, wherever the ring ends
        // run predecessors still inside the ring (4094 item starts of the context back) / within 510 item starts
        const DistBracket d5 = dist_valid(codes, p > h5 ? p - h5 : 0);
        uint32_t v4 = count_from(d4, r, j, h4);
        if (a.dbg & 2) v4 = 64;
        const uint32_t span = fast_min(r, v4);
        if (span < 64) mask = span ? mask & (~0ull << (64 - span)) : 0;
        // the candidates, newest first, straight from the registers: no load in this loop
        uint32_t best = 0, bk = 0, m1 = 0, m2 = 0, seen = 0;
        bool full = false;  // a candidate matched all 240 bytes: nothing older is looked at
#pragma unroll
        for (uint32_t k = 0; k < kFastK; k++) {
            if ((k & 15) == 0 && !(mask & (~0ull >> k))) break;  // no item start left in the window
            const uint32_t l = (rw[k >> 2] >> (8 * (k & 3))) & 0xff;
            if (((mask >> (63 - k)) & 1) && !full && seen < a.depth) {
                if (l > best) { best = l; bk = k; }
                if (seen < a.lazy1 && l > m1) m1 = l;
                if (seen < a.lazy2 && l > m2) m2 = l;
                seen++;
                full = l == kMaxLen;
            }
        }
        const bool stop = full || v4 < r;  // (the ring ends inside the window: nothing older counts either)
        uint32_t b510 = best && (bk < d5.sure || (bk < d5.limit && a.epos[j - 1 - bk] >= h5));
        if (!stop && longrun) {
            uint32_t fv;
            if (scan) {
                // (Tried in round 5 and dropped: the scans as a kernel of their own over a densely packed list of requests -- FastEval
                // without them 65 us at eight waves a SIMD, the scan kernel 91 us: together more than the 100 us of this kernel,
                // whose window-only wavefronts run beside its scanning ones.)
                fv = scan_far(p, i, c, h4, h5, codes, j, r, wbits, seen, rnd, first);
                a.farv[i] = fv;

            } else {
                fv = first ? 0 : a.farv[i];
            }
            if (fv >> 25) {  // merge (these candidates are older than every one of the window: they win only when strictly longer)
                if ((fv & 0xff) > best) { best = fv & 0xff; b510 = (fv >> 24) & 1; }
                if (((fv >> 8) & 0xff) > m1) m1 = (fv >> 8) & 0xff;
                if (((fv >> 16) & 0xff) > m2) m2 = (fv >> 16) & 0xff;
            }
        } else if (scan) {
            a.farv[i] = 0;  // (nothing beyond the window counts now; nothing is remembered)
        }
        if ((a.dbg & (128 | 256)) && first && !scan) a.farv[i] = 0;  // (experiments without a first-round scan: nothing is remembered yet)
        // word predictor (src/lz.rs:132-133): newest update u <= p-2 with hash2(u-1) == hash2(p-1)
        if (rk < 64) kmask = rk ? kmask & (~0ull << (64 - rk)) : 0;
        if (km & 0x80) kmask &= ~(1ull << 63);
        const uint32_t lwm = kmask ? (uint32_t)((wm >> (63 - (uint32_t)clz64(kmask))) & 1) : (km >> 8) & 1;
#if !defined(__HIPCC__) && !defined(ORZ_EMU_THREADS)
        if ((a.dbg & 32) && p < r2lo && !a.dirty[i] && !moved && !scan) {  // verify mode: would a skipped position have changed?
            const uint32_t o = a.ev[i], nw = best | (m1 << 8) | (m2 << 16) | (lwm << 24) | (b510 << 25);
            if ((o ^ nw) & ~(1u << 25)) {
                g_eval_stats[6]++;
                if ((o ^ nw) & (1u << 24)) g_eval_stats[7]++;
                static int shown = 0;
                if (shown++ < 20) fprintf(stderr, "skip-miss p=%u old=%08x new=%08x rl=%u\n", p, o, nw, rl);
            }
        }
#endif
        // ---- stores last: a load queued behind scattered stores would wait for them
        a.ev[i] = best | (m1 << 8) | (m2 << 16) | (lwm << 24) | (b510 << 25);
        if (dirty && !(a.dbg & 32)) a.dirty[i] = 0;
    }
};
// number of set bits among the `count` bits of a bitmap that start at bit `start`
ORZ_D uint32_t popc_range(const uint64_t* bm, uint32_t start, uint32_t count) {
    uint32_t n = 0;
    while (count) {
        const uint32_t sh = start & 63, take = fast_min(count, 64 - sh);
        uint64_t w = bm[start >> 6] >> sh;
        if (take < 64) w &= (1ull << take) - 1;
        n += (uint32_t)popc64(w);
        start += take; count -= take;
    }
    return n;
}
// A tile has had its last round: its item starts are final (its entry comes from tiles that were final before it).  Their
// records go to the compact lists, behind what each run holds already, in position order: a run's slots ascend with the
// position, so an item start's place among the tile's members of its run is the number of set bits between the run's
// first slot inside the tile and its own (found by galloping down the slot positions; one or two loads on text).  The
// counters move in a second launch (FastRetireDone), so every member of a run reads the same base: no order depends on
// the scheduling of the threads.
struct FastRetire {  // thread per position of the tile
    FastArgs a;
    uint32_t lo, hi;  // the tile, window offsets [lo, hi)
    uint32_t* place;  // [n] 1 + place of each appended member among the tile's members of its run (cleared by FastRetireDone)
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t p = lo + (uint32_t)tid;
        if (p >= hi) return;
        const uint32_t i = p - kPre;
        if (!a.mfb[i]) return;
        const uint32_t j = a.idx[p], key = bucket_key(a.win, p), rl = a.rlen[i];
        const uint32_t rs = a.runstart[key], base = a.ccnt[key];
        const uint64_t t0 = a.stext[2 * (size_t)j], t1 = a.stext[2 * (size_t)j + 1];  // (asked for before the walk down the slots)
        const uint32_t maxk = rl < 255 ? rl : j - rs;  // run slots below j
        uint32_t good = 0, bad = maxk + 1;             // slots j - 1 .. j - good hold positions of this tile, slot j - bad does not
        for (uint32_t k = 1; k <= maxk; k <<= 1) {
            if (a.epos[j - k] >= lo) good = k; else { bad = k; break; }
        }
        while (good + 1 < bad) {
            const uint32_t mid = (good + bad) / 2;
            if (a.epos[j - mid] >= lo) good = mid; else bad = mid;
        }
        const uint32_t at = popc_range(a.vbits, j - good, good);
        uint64_t* dst = a.cl + 2 * ((size_t)rs + base + at);
        dst[0] = t0; dst[1] = t1;
        if (!(a.dbg & 1024)) atom_add32(&a.cnew[key], 1);  // (experiments: 1024 = the time of the kernel without this atomic)
        place[i] = at + 1;
    }
};
struct FastRetireDone {  // the first member of each (run, tile) group moves the run's counter
    FastArgs a;
    uint32_t lo, hi;
    uint32_t* place;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t p = lo + (uint32_t)tid;
        if (p >= hi) return;
        const uint32_t v = place[p - kPre];
        if (!v) return;
        place[p - kPre] = 0;
        if (v != 1) return;
        const uint32_t key = bucket_key(a.win, p);
        a.ccnt[key] += a.cnew[key];
        a.cnew[key] = 0;
    }
};
struct FastListReset {  // a block parsed again (finer tiles): the lists go back to their history heads (thread per slot)
    const uint32_t *epos, *keys, *runstart;  // (the slots' sorted keys: the last three history positions were filed under the keys
    uint32_t nent;                           // they had before the slide, BuildKeys)
    uint32_t* ccnt;  // (zeroed)
    ORZ_HD void operator()(size_t j) const {
        if (j >= nent) return;
        if (epos[j] < kPre && (j + 1 == nent || keys[j + 1] != keys[j] || epos[j + 1] >= kPre)) ccnt[keys[j]] = (uint32_t)j + 1 - runstart[keys[j]];
    }
};

// Ring horizons of the subtiles [s0, s1] per context: the oldest window offset from which at most `limit` item starts of
// the context lie before the END of the subtile -- candidates at or after it are inside the ring (limit 4094 - margin) /
// cost fewer than 8 offset bits (510), whatever the position inside the subtile (conservative by the subtile's own
// count).  Counts come from cp (new region, kept by FastPrefix; the tile about to start is extrapolated there) and hpre
// (history).  Thread per (subtile, ctx), two binary searches over L2-resident columns.
struct FastHorizon {
    FastArgs a;
    uint32_t s0, s1;
    ORZ_D uint32_t horizon(uint32_t s, uint32_t c, uint32_t limit) const {
        // counted from the MIDDLE of the subtile's own item starts (a position sees those before it: half of them on average)
        const uint32_t c0 = a.cp[(size_t)s * 256 + c], c1 = a.cp[(size_t)(s + 1) * 256 + c];
        const uint32_t end = c0 + (c1 - c0) / 2, base = a.cp[c];
        if (end - base > limit) {  // inside the new region: smallest s' in [0, s + 1] with end - cp[s'] <= limit
            uint32_t lo = 0, hi = s + 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) / 2;
                const uint32_t v = a.cp[(size_t)mid * 256 + c];
                if (v >= end || end - v <= limit) hi = mid; else lo = mid + 1;
            }
            uint32_t h = kPre + lo * kSub;
            if (lo) {  // part of the subtile before still fits: its item starts are taken as evenly spread
                const uint32_t v = a.cp[(size_t)lo * 256 + c], room = limit - (v >= end ? 0 : end - v);
                const uint32_t m = v - a.cp[(size_t)(lo - 1) * 256 + c];
                if (m) h -= (uint32_t)(((uint64_t)fast_min(room, m) * kSub) / m);
            }
            return h;
        }
        const uint32_t room = limit - (end - base), tot = a.hpre[(size_t)kHistSub * 256 + c];
        uint32_t lo = 0, hi = kHistSub;  // smallest unified subtile u with tot - hpre[u] <= room
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (tot - a.hpre[(size_t)mid * 256 + c] <= room) hi = mid; else lo = mid + 1;
        }
        if (!lo) return 1;
        uint32_t h = lo * kSub - 1;
        const uint32_t left = room - (tot - a.hpre[(size_t)lo * 256 + c]);
        const uint32_t m = a.hpre[(size_t)lo * 256 + c] - a.hpre[(size_t)(lo - 1) * 256 + c];
        if (m) h -= (uint32_t)(((uint64_t)fast_min(left, m) * kSub) / m);
        return h > 1 ? h : 1;
    }
    // a thread per (subtile, ctx, which horizon): the two binary searches side by side instead of one after the other
    ORZ_HD size_t threads() const { return (size_t)(s1 - s0 + 1) * 512; }
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t which = (uint32_t)(tid & 1), c = (uint32_t)((tid >> 1) & 255), s = s0 + (uint32_t)(tid >> 9);
        if (s > s1) return;
        uint32_t* hz = a.hz + ((size_t)s * 256 + c) * 4;
        hz[2 + which] = hz[which];
        hz[which] = horizon(s, c, which ? 510 : kRing - kRingMargin);
    }
};
// src/lz.rs:139-234 on the snapshot's answers: e = ev of the position, e1 / e2 = ev of the next two (0 past the end);
// returns type | advance << 8
ORZ_D uint32_t fast_decide(uint32_t p, uint32_t len, uint32_t e, uint32_t e1, uint32_t e2) {
    // (selects, no early returns: PathUpWave makes 64 of these decisions a lane and the branchy form was a branch per rule)
    uint32_t L = e & 0xff;
    L = p + L >= len ? len - 1 - p : L;  // an item never reaches the block end (src/matcher.rs:183)
    L = L < kMinLen ? 0 : L;
    const uint32_t lwm = (e >> 24) & 1;
    const uint32_t l1 = L + 1 + ((e >> 25) & 1), l2 = l1 - lwm;
    const bool look = L > 0 && L < kMaxLen / 2;
    const bool lazy1 = look && ((e1 >> 8) & 0xff) >= l1;
    const bool lazy2 = look && !lazy1 && ((e2 >> 16) & 0xff) >= l2;
    const bool match = L > 0 && !lazy1 && !lazy2;
    const bool word = !match && p + 1 < len && !lazy1 && lwm != 0;
    return match ? (kTyMatch | (L << 8)) : (word ? (kTyWord | (2u << 8)) : (kTyLit | (1u << 8)));
}
struct FastDecide {  // thread per position (measured: inside PathUpWave the four wavefronts of a chunk each pay for it, +20 us a step)
    FastArgs a;
    uint32_t lo, hi;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t p = lo + (uint32_t)tid;
        if (p >= hi) return;
        const uint32_t i = p - kPre;
        const uint32_t d = fast_decide(p, a.len, a.ev[i], p + 1 < a.len ? a.ev[i + 1] : 0, p + 2 < a.len ? a.ev[i + 2] : 0);
        a.ty[i] = (uint8_t)d; a.nl[i] = (uint8_t)(d >> 8);
    }
};

// ---- path through the active range: exit maps per segment / chunk / tile, then entries top-down --------
ORZ_D uint32_t seg_end(uint32_t s, uint32_t len) { return fast_min(len, kPre + (s + 1) * kSeg64); }
ORZ_D uint32_t chunk_end(uint32_t c, uint32_t len) { return fast_min(len, kPre + (c + 1) * kSub); }
struct PathSeg {  // x0[p] = where a walk from p first leaves p's 64-position segment, as an offset past its end
    FastArgs a;
    uint32_t lo, hi;
    ORZ_HD void operator()(size_t tid) const {
        uint32_t x = lo + (uint32_t)tid;
        if (x >= hi) return;
        const uint32_t i = x - kPre, e = seg_end(i / kSeg64, a.len);
        while (x < e) { const uint32_t d = a.nl[x - kPre]; x += d ? d : 1; }
        a.x0[i] = (uint8_t)(x - e);
    }
};
struct PathChunk {  // x1[c][e] = exit offset past chunk c when entering it at offset e
    FastArgs a;
    uint32_t c0, nc;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t c = c0 + (uint32_t)(tid / kEntries), e = (uint32_t)(tid % kEntries);
        if (c >= c0 + nc) return;
        const uint32_t end = chunk_end(c, a.len);
        uint32_t x = kPre + c * kSub + e;
        while (x < end) {
            const uint32_t s = (x - kPre) / kSeg64;
            x = seg_end(s, a.len) + a.x0[x - kPre];
        }
        a.x1[(size_t)c * kEntries + e] = (uint8_t)(x >= end ? x - end : 0);
    }
};
// The same two maps, one wavefront per 4096-position chunk with the advance lengths staged in LDS: lane s
// fills x0 for segment s by a backward sweep (a position either leaves the segment or inherits the exit of
// the position it jumps to), then the 240 chunk entries are walked through LDS.  Rows are padded to 68 bytes
// so that the 64 lanes, which walk 64 different segments in lockstep, hit different banks.
ORZ_D uint32_t pad68(uint32_t x) { return (x >> 6) * 68 + (x & 63); }
// eight consecutive bytes (x a multiple of 8) into a padded table: two 32-bit LDS stores (a row of 64 starts at a multiple of 4)
ORZ_D void st8_pad68(uint8_t* tab, uint32_t x, uint64_t v) {
    uint32_t* d = reinterpret_cast<uint32_t*>(tab + pad68(x));
    d[0] = (uint32_t)v; d[1] = (uint32_t)(v >> 32);
}
struct PathUpWave {
    FastArgs a;
    uint32_t c0;
    // round 5: the launch can also DECIDE (FastDecide's job: every part makes the chunk's decisions from ev for its own staging --
    // four times the arithmetic, one launch less -- and stores its quarter of ty / nl) and fold the counters of the tile that
    // retired in the step before (FastRetireDone: the blocks behind the chunks' 4 x nchunks).  Alone a step is 20 us longer that
    // way (round 2's measurement); under eight encoders a launch waits ~100 us for its turn whatever its size (rocprofv3, eight encoders: FastDecide 4.9 -> 124 us,
    // FastRetireDone 6.5 -> 125 us, a fill 4.5 -> 118 us): the number of launches a block is what the aggregate pays for.
    uint32_t decide = 0, nup = 0;  // decide: 0 = advances come from a.nl; nup: blocks that belong to the chunks
    FastRetireDone done{};
    static size_t lds_bytes() { return 2 * 64 * 68 + 64; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        if (decide && w.block() >= nup) {  // the piggy-backed FastRetireDone
            const uint32_t t = (w.block() - nup) * 64 + w.lane();
            done((size_t)t);
            return;
        }
        uint8_t* nlL = w.lds();
        uint8_t* x0L = nlL + 64 * 68 + 32;
        const uint32_t c = c0 + w.block() / 4, part = w.block() & 3, lane = w.lane();  // four wavefronts per chunk: 60 entries each
        const uint32_t cs = kPre + c * kSub, clen = chunk_end(c, a.len) - cs;
        // (512 words of 8 positions, 8 per lane.  Round 6: the lane's forty loads of ev go out together -- a chunk's wavefront has
        // its SIMD to itself, nothing else hides a load's latency, and eight trips of load-then-decide were eight round trips.)
        uint64_t w5[8][5];
        if (decide) {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = (k * 64 + lane) * 8;
                const uint64_t* ew = reinterpret_cast<const uint64_t*>(a.ev + (cs - kPre) + (x < clen ? x : 0));
#pragma unroll
                for (uint32_t q = 0; q < 5; q++) w5[k][q] = ew[q];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t x = (k * 64 + lane) * 8;
            if (!decide) {
                st8_pad68(nlL, x, x < clen ? *reinterpret_cast<const uint64_t*>(a.nl + (cs - kPre) + x) : 0);
                continue;
            }
            uint64_t nlw = 0, tyw = 0;
            if (x < clen) {
                const uint32_t i0 = (cs - kPre) + x;  // (a multiple of 8: the ten answers the eight decisions look at; ev holds 512
                uint32_t e[10];                       // entries more than the block, the answers past the block end are zeroed here)
#pragma unroll
                for (uint32_t q = 0; q < 10; q++) {
                    const uint32_t v = (uint32_t)(w5[k][q >> 1] >> (32 * (q & 1)));
                    e[q] = cs + x + q < a.len ? v : 0;
                }
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) {
                    const uint32_t p = cs + x + q;
                    const uint32_t d = p < a.len ? fast_decide(p, a.len, e[q], e[q + 1], e[q + 2]) : 0;
                    tyw |= (uint64_t)(d & 0xff) << (8 * q);
                    nlw |= (uint64_t)((d >> 8) & 0xff) << (8 * q);
                }
                if ((x >> 10) == part) {  // every part stores a quarter (positions < clen only: a chunk's tail word is cut below)
                    if (x + 8 <= clen) {
                        *reinterpret_cast<uint64_t*>(a.ty + i0) = tyw;
                        *reinterpret_cast<uint64_t*>(a.nl + i0) = nlw;
                    } else {
                        for (uint32_t q = 0; x + q < clen; q++) { a.ty[i0 + q] = (uint8_t)(tyw >> (8 * q)); a.nl[i0 + q] = (uint8_t)(nlw >> (8 * q)); }
                    }
                }
            }
            st8_pad68(nlL, x, nlw);
        }
        w.sync();
        // Exits of the lane's own segment by a backward sweep (a position either leaves the segment or inherits the exit of the
        // position it jumps to).  The lane's 64 advances come out of LDS in sixteen independent reads first, so that a step of
        // the sweep is ONE dependent LDS round trip (the inherited exit) instead of two.  (Tried and dropped: pointer jumping with
        // lane = position, all segments a round -- 49 us against 21: every hop waits for the store before it.)
        {
            const uint32_t s0 = lane * 64;
            if (s0 < clen) {
                const uint32_t s1 = fast_min(s0 + 64, clen), sl = s1 - s0;
                uint32_t adv[16];
                const uint32_t* row = reinterpret_cast<const uint32_t*>(nlL + lane * 68);
#pragma unroll
                for (uint32_t q = 0; q < 16; q++) adv[q] = row[q];
                uint8_t* xrow = x0L + lane * 68;
#pragma unroll
                for (uint32_t k = 0; k < 64; k++) {
                    const uint32_t p = 63 - k;
                    if (p < sl) {
                        const uint32_t d = (adv[p >> 2] >> (8 * (p & 3))) & 0xff, x = p + (d ? d : 1);
                        xrow[p] = x >= sl ? (uint8_t)(x - sl) : xrow[x];
                    }
                }
            }
        }
        w.sync();
        for (uint32_t k = part * 2; k < part * 2 + 2; k++) {  // every part writes a quarter of x0
            const uint32_t x = (k * 64 + lane) * 8;
            if (x >= clen) continue;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(x0L + pad68(x));
            *reinterpret_cast<uint64_t*>(a.x0 + (cs - kPre) + x) = (uint64_t)src[0] | ((uint64_t)src[1] << 32);
        }
        w.sync();
        if (lane < 60) {
            const uint32_t e = part * 60 + lane;
            uint32_t x = e;
            while (x < clen) {
                const uint32_t s1 = fast_min(((x >> 6) + 1) * 64, clen);
                x = s1 + x0L[pad68(x)];
            }
            a.x1[(size_t)c * kEntries + e] = (uint8_t)(x - clen);
        }
    }
};
struct PathMarkWave {  // one wavefront per chunk, lane = segment; also the chunk's item starts per ctx (what CountWave counts)
    FastArgs a;
    uint32_t c0;
    static constexpr uint32_t kTab = 64 * 68 + 32;
    static size_t lds_bytes() { return 3 * kTab + (65 * 68 + 12) + 256 * 4; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        uint8_t* nlL = w.lds();
        uint8_t* x0L = nlL + kTab;
        uint8_t* tyL = x0L + kTab;
        uint8_t* wL = tyL + kTab;  // the window's bytes from two before the chunk on: the ctx of an item start without a load from memory inside the walk
        uint32_t* cnt = (uint32_t*)(wL + 65 * 68 + 12);
        const uint32_t c = c0 + w.block(), lane = w.lane();
        const uint32_t cs = kPre + c * kSub, clen = chunk_end(c, a.len) - cs;
        for (uint32_t k = lane; k < 256; k += 64) cnt[k] = 0;
        {   // (the lane's 32 loads in flight together, then the stores: see PathUpWave)
            uint64_t v[8], u[8], t[8], b[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = (k * 64 + lane) * 8;
                v[k] = u[k] = t[k] = 0;
                if (x < clen) {
                    v[k] = *reinterpret_cast<const uint64_t*>(a.nl + (cs - kPre) + x);
                    u[k] = *reinterpret_cast<const uint64_t*>(a.x0 + (cs - kPre) + x);
                    t[k] = *reinterpret_cast<const uint64_t*>(a.ty + (cs - kPre) + x);
                }
                b[k] = ldu64(a.win + cs - 2 + x);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = (k * 64 + lane) * 8;
                st8_pad68(nlL, x, v[k]); st8_pad68(x0L, x, u[k]); st8_pad68(tyL, x, t[k]);
                st8_pad68(wL, x, b[k]);  // wL[pad68(y)] = win[cs - 2 + y]
            }
        }
        if (lane == 0) st8_pad68(wL, kSub, ldu64(a.win + cs - 2 + kSub));
        w.sync();
        const uint32_t s0 = lane * 64;
        if (s0 < clen) {
            const uint32_t s1 = fast_min(s0 + 64, clen);
            const uint32_t ce = a.centry[c];
            uint32_t x = ce >= cs ? ce - cs : 0;
            while (x < s0) {
                const uint32_t e1 = fast_min(((x >> 6) + 1) * 64, clen);
                x = e1 + x0L[pad68(x)];
            }
            uint64_t m = 0;
            while (x < s1) {
                m |= 1ull << (x - s0);
                atom_add32(&cnt[(uint32_t)(wL[pad68(x + 1)] & 0x7f) | ((uint32_t)is_alnum(wL[pad68(x)]) << 7)], 1);
                const uint32_t d = nlL[pad68(x)];
                const uint32_t e = x + (d ? d : 1);
                a.pt[(cs - kPre) + e] = tyL[pad68(x)];
                x = e;
            }
            a.sbits[(cs - kPre) / 64 + lane] = m;
        }
        w.sync();
        for (uint32_t k = lane; k < 256; k += 64) a.cm[(size_t)c * 256 + k] = cnt[k];
    }
};
ORZ_D uint32_t tile_end(uint32_t t, uint32_t tile, uint32_t len) { return fast_min(len, kPre + (t + 1) * tile); }
// Per (tile, entry) exits, then the entries top-down (tile -> chunk), in ONE launch of one workgroup: the chunk maps of the
// range are staged in LDS first (a walk is a chain of dependent loads: ~32 per tile entry, ~35 per chunk entry), the
// tile maps stay there too.  Three phases around two barriers (backend launch_group).
struct PathTileDown {
    FastArgs a;
    uint32_t t0, nt;
    static constexpr uint32_t kPhases = 3;
    static constexpr size_t kLdsMax = 144 * 1024;  // (of the CU's 160 KB; three rounds of 512 KiB tiles need 93 KB)
    ORZ_HD uint32_t range_chunks() const {
        const uint32_t cpt = a.tile / kSub, c0 = t0 * cpt;
        const uint32_t end = kPre + (t0 + nt) * a.tile;
        return ((a.len < end ? a.len : end) - (kPre + c0 * kSub) + kSub - 1) / kSub;
    }
    size_t lds_bytes() const {  // 0 = the range's maps do not fit: walk them in global memory
        const size_t need = ((size_t)range_chunks() + nt) * kEntries + 16;
        return need <= kLdsMax ? need : 0;
    }
    ORZ_HD void phase(uint32_t ph, uint32_t tid, uint32_t nth, uint8_t* lds, bool use_lds) const {
        const uint32_t cpt = a.tile / kSub, c0 = t0 * cpt, nch = range_chunks();
        uint8_t* x2L = lds + (((size_t)nch * kEntries + 15) & ~(size_t)15);
        // maps of chunk c0 / tile t0 onwards: the staged copies or the arrays themselves (indexed relative to c0 / t0; a
        // biased LDS pointer would leave the LDS aperture as a flat address)
        const uint8_t* x1p = use_lds ? lds : a.x1 + (size_t)c0 * kEntries;
        uint8_t* x2p = use_lds ? x2L : a.x2 + (size_t)t0 * kEntries;
        if (ph == 0) {
            if (!use_lds) return;
            const uint8_t* src = a.x1 + (size_t)c0 * kEntries;  // (c0 * 240 is a multiple of 16)
            // (round 6: sixteen bytes a thread and four of those loads in flight -- ONE workgroup stages up to 90 KB, and a word a
            // thread a trip was a chain of twenty round trips to memory)
            struct alignas(16) Q { uint32_t w[4]; };
            const uint32_t nq = nch * kEntries / 16;
            const Q* s16 = reinterpret_cast<const Q*>(src);
            Q* d16 = reinterpret_cast<Q*>(lds);
            for (uint32_t k0 = tid; k0 < nq; k0 += 4 * nth) {
                Q v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) if (k0 + u * nth < nq) v[u] = s16[k0 + u * nth];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) if (k0 + u * nth < nq) d16[k0 + u * nth] = v[u];
            }
        } else if (ph == 1) {
            for (uint32_t k = tid; k < nt * kEntries; k += nth) {
                const uint32_t t = t0 + k / kEntries, e = k % kEntries;
                const uint32_t end = tile_end(t, a.tile, a.len);
                uint32_t x = kPre + t * a.tile + e;
                while (x < end) {
                    const uint32_t c = (x - kPre) / kSub;
                    x = chunk_end(c, a.len) + x1p[(size_t)(c - c0) * kEntries + (x - kPre - c * kSub)];
                }
                const uint8_t v = (uint8_t)(x >= end ? x - end : 0);
                a.x2[(size_t)t * kEntries + e] = v;
                if (use_lds) x2p[(size_t)(t - t0) * kEntries + e] = v;
            }
        } else {
            for (uint32_t k = tid; k <= nch; k += nth) {  // chunk k of the range; k == nch: the entry of the tile after the range
                const bool tail = k == nch;
                const uint32_t c = c0 + k, t = tail ? t0 + nt : c / cpt;
                uint32_t x = a.tentry[t0];
                for (uint32_t tt = t0; tt < t; tt++) {
                    const uint32_t end = tile_end(tt, a.tile, a.len);
                    if (x < end) x = end + x2p[(size_t)(tt - t0) * kEntries + (x - kPre - tt * a.tile)];
                }
                if (tail) { a.tentry[t] = x; continue; }
                if (c == t * cpt && t > t0) a.tentry[t] = x;
                for (uint32_t cc = t * cpt; cc < c; cc++) {
                    const uint32_t end = chunk_end(cc, a.len);
                    if (x < end) x = end + x1p[(size_t)(cc - c0) * kEntries + (x - kPre - cc * kSub)];
                }
                a.centry[c] = x;
            }
        }
    }
};
struct PathMark {  // thread per segment: its item starts as a 64-bit mask; the type of every item at its end
    FastArgs a;
    uint32_t s0, ns;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t s = s0 + (uint32_t)tid;
        if (tid >= ns) return;
        const uint32_t start = kPre + s * kSeg64, end = seg_end(s, a.len);
        uint32_t x = a.centry[(start - kPre) / kSub];
        while (x < start) {
            const uint32_t ss = (x - kPre) / kSeg64;
            x = seg_end(ss, a.len) + a.x0[x - kPre];
        }
        uint64_t m = 0;
        while (x < end) {
            m |= 1ull << (x - start);
            const uint32_t d = a.nl[x - kPre];
            const uint32_t e = x + (d ? d : 1);
            a.pt[e - kPre] = a.ty[x - kPre];
            x = e;
        }
        a.sbits[s] = m;
    }
};
// Bring the slot-order bitmaps in line with the path (thread per position), and mark the positions whose next evaluation
// would see the difference (dirty).  A flipped item start matters to the later positions of its run for which it is among
// the newest `dmax` item starts below them: walk up the run (slots ascend with the position) until that many set bits have
// been passed, the run ends or the range that is still being re-evaluated (< mark_hi) is left; a flipped word update
// matters up to the next set bit above it.  Bits that flip concurrently are covered by their own walks: whichever state
// a walk observes, the union of the marks contains every position whose answer can have changed.
// A thread is one short chain: its loads first (the walks only COUNT the slots to mark, sixteen slots per trip), then
// its stores and atomics -- on this hardware a load queued behind scattered stores waits for them.  All atomics are
// fire-and-forget: the summary level v1 is only ever set here (a bit whose word went back to zero costs a walker one
// wasted load; V1Build makes it exact again at the start of a parse).
struct FastFlip {
    FastArgs a;
    uint32_t lo, hi;     // positions y in [lo, hi]
    uint32_t next_entry; // tile index whose entry position also counts as an item start (or ~0u)
    uint32_t mark_hi;    // positions below this one are marked dirty: those beyond are evaluated in the next step whatever their
                         // flags say (FastEval r2lo); 0 = no marking (the repair passes)
    uint32_t last_hi;    // positions below this one are in their tile's last round: item starts that still change there are counted
    uint32_t* lastflips;
    // slots above `slot` whose positions are to be marked
    static constexpr uint32_t kTrip = 16;  // slots per trip: a walk is at most four rounds of independent loads
    ORZ_D uint32_t walk(bool words, uint32_t slot, uint32_t y) const {
        const uint32_t* pos = words ? a.kpos : a.epos;
        const uint64_t* bits = words ? a.kbits : a.vbits;
        const uint32_t lim = words ? 1u : a.dmax;
        const uint32_t end = words ? fast_min(a.nk, slot + 1 + 64) : fast_min(*a.nentp, slot + 1 + kFastK);  // (the windows that show the slot)
        uint32_t passed = 0, n = 0;
        for (uint32_t s = slot + 1; s < end; s += kTrip) {
            if (a.dbg & 64) atom_add64(&a.stats[words ? 19 : 18], 1);
            uint32_t q[kTrip];
#pragma unroll
            for (uint32_t b = 0; b < kTrip; b++) q[b] = s + b < end ? pos[s + b] : 0;
            // (other threads of this launch flip bits of these words: whichever state a walk sees, the union of the marks covers
            // every position whose answer can have changed -- see the kernel's comment)
            const uint64_t w0 = racy_load64(&bits[s >> 6]), w1 = racy_load64(&bits[(s + kTrip - 1) >> 6]);
#pragma unroll
            for (uint32_t b = 0; b < kTrip; b++) {
                const uint32_t qq = q[b];
                if (s + b >= end || qq <= y || qq >= mark_hi) return n;  // another run / evaluated anyway next step
                n++;
                const uint64_t w = ((s + b) >> 6) == (s >> 6) ? w0 : w1;
                if (((w >> ((s + b) & 63)) & 1) && ++passed >= lim) return n;
            }
        }
        return n;
    }
    ORZ_D void mark(bool words, uint32_t slot, uint32_t n) const {
        const uint32_t* pos = words ? a.kpos : a.epos;
        for (uint32_t k = 0; k < n; k += kTrip) {
            uint32_t q[kTrip];
#pragma unroll
            for (uint32_t b = 0; b < kTrip; b++) q[b] = k + b < n ? pos[slot + 1 + k + b] : 0;
#pragma unroll
            for (uint32_t b = 0; b < kTrip; b++)
                if (k + b < n && q[b] >= kPre) racy_store8(&a.dirty[q[b] - kPre], 1);  // (several walks may mark one position: the same value)
        }
    }
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t y = lo + (uint32_t)tid;
        if (y > hi) return;
        flip_one(y, nullptr);
    }
    // Which of the eight positions lo + t .. lo + t + 7 (t a multiple of 8, lo - kPre a multiple of 8) have anything to flip: the
    // tests of flip_one on 8-byte loads of the flag arrays -- most positions of a step have nothing to do, and a thread per
    // position that finds that out costs a wavefront per 64 of them (FlipPrefixWave).
    ORZ_D uint32_t flip_mask8(uint32_t t) const {
        const uint32_t y0 = lo + t, i0 = y0 - kPre;
        if (y0 > hi) return 0;
        const uint32_t exit_at = next_entry != ~0u ? a.tentry[next_entry] : a.len;
        const uint64_t mf8 = *reinterpret_cast<const uint64_t*>(a.mfb + i0), ef8 = *reinterpret_cast<const uint64_t*>(a.efb + i0);
        const uint64_t pt8 = *reinterpret_cast<const uint64_t*>(a.pt + i0);
        const uint32_t sb = (uint32_t)((a.sbits[i0 / 64] >> (i0 & 63)) & 0xff);
        uint32_t m = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t y = y0 + k;
            if (y > hi) break;
            const uint32_t mf = (uint32_t)(mf8 >> (8 * k)) & 0xff, ef = (uint32_t)(ef8 >> (8 * k)) & 0xff, ptk = (uint32_t)(pt8 >> (8 * k)) & 0xff;
            const uint32_t s = (uint32_t)(y == exit_at) | (y < a.len ? (sb >> k) & 1 : 0);
            const uint32_t sw = y < a.len ? s : mf;
            const uint32_t ew = y >= kPre + 1 ? (uint32_t)(s && ptk != kTyWord) : ef;
            if (sw != mf || ew != ef) m |= 1u << k;
        }
        return m;
    }
    // kdirty (repair stage): one bit per hash2 key whose word-update bits changed -- the WORD items of those keys are judged again
    ORZ_D void flip_one(uint32_t y, uint64_t* kdirty) const {
        const uint32_t i = y - kPre;
        const uint32_t exit_at = next_entry != ~0u ? a.tentry[next_entry] : a.len;  // where the path leaves the range / the block
        const uint32_t mf = a.mfb[i], ef = a.efb[i];
        const uint32_t s = (uint32_t)(y == exit_at) | (y < a.len ? (uint32_t)((a.sbits[i / 64] >> (i & 63)) & 1) : 0);
        const uint32_t sw = y < a.len ? s : mf;
        // words[] update of the item ending at y (src/lz.rs:203,233): u = y - 2
        const uint32_t ew = y >= kPre + 1 ? (uint32_t)(s && a.pt[i] != kTyWord) : ef;
        const bool dv = sw != mf, de = ew != ef;
        if (!dv && !de) return;
        // ---- loads
        const uint32_t j = dv ? a.idx[y] : 0, ku = de ? a.kidx[y - 2] : 0;
        const bool nowalk = (a.dbg & 512) != 0;  // (experiments: the time of the kernel without its walks)
        const uint32_t nv = dv && y < mark_hi && !nowalk ? walk(false, j, y) : 0;
        const uint32_t nk = de && y - 2 < mark_hi && !nowalk ? walk(true, ku, y - 2) : 0;
        // ---- stores
        if (dv) {
            if (y < last_hi) atom_add32(lastflips, 1);
            if (a.dbg & 64) atom_add64(&a.stats[16], 1);
            // (the summary bit only has to be set by the flip that makes an empty word non-empty: the XOR hands back the word it found)
            const uint64_t was = atom_fetch_xor64(&a.vbits[j >> 6], 1ull << (j & 63));
            if (sw && was == 0) atom_or64(&a.v1[j >> 12], 1ull << ((j >> 6) & 63));
            a.mfb[i] = (uint8_t)sw;
            mark(false, j, nv);
        }
        if (de) {
            if (a.dbg & 64) atom_add64(&a.stats[17], 1);
            const uint64_t kwas = atom_fetch_xor64(&a.kbits[ku >> 6], 1ull << (ku & 63));
            if (ew && kwas == 0) atom_or64(&a.k1[ku >> 12], 1ull << ((ku >> 6) & 63));
            a.efb[i] = (uint8_t)ew;
            mark(true, ku, nk);
            if (kdirty) { const uint32_t key2 = hash2(a.win, y - 3); atom_or64(&kdirty[key2 >> 6], 1ull << (key2 & 63)); }
        }
    }
};
// item starts per (subtile, ctx) of the current path: one wavefront per 4096-position subtile, lane = 64 positions,
// counters in LDS (the ring ordinals of a round are estimated from these: FastEval)
struct CountWave {
    FastArgs a;
    uint32_t s0;
    static size_t lds_bytes() { return 256 * 4; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        uint32_t* cnt = (uint32_t*)w.lds();
        const uint32_t s = s0 + w.block(), lane = w.lane();
        for (uint32_t c = lane; c < 256; c += 64) cnt[c] = 0;
        w.sync();
        const uint32_t i0 = s * kSub + lane * 64;
        if (i0 < a.n) {
            uint64_t m = a.sbits[i0 / 64];
            const uint8_t* b = a.win + kPre + i0;
            while (m) {
                const uint32_t t = (uint32_t)ctz64(m);
                m &= m - 1;
                if (i0 + t < a.n) atom_add32(&cnt[(uint32_t)(b[(int)t - 1] & 0x7f) | ((uint32_t)is_alnum(b[(int)t - 2]) << 7)], 1);
            }
        }
        w.sync();
        for (uint32_t c = lane; c < 256; c += 64) a.cm[(size_t)s * 256 + c] = cnt[c];
    }
};
// cp[s][c] = cp[s0][c] + sum of counts[s0 .. s)[c] for s in (s0, s1 + ext].
// The counts are the path's own (cm) for the subtiles below `live_end` -- the tile in its last round, whose path is final
// once this step has run -- and for the tiles behind it the counts of the same places in the newest tile below live_end:
// the ring horizons of tiles that are still settling must not follow their own unsettled item counts (a sketch with too
// many items pulls the horizon in, fewer candidates count, the next round makes too many items again: zeros with noise
// came out 6 % larger that way).  While no tile is in its last round yet (the first steps) live_end = s0: nothing is
// counted, the horizons lie in the history.  The `ext` subtiles behind s1 belong to the tile that starts next.
struct FastPrefix {
    FastArgs a;
    uint32_t s0, s1, ext, cpt;
    uint32_t live_end;
    ORZ_D uint32_t cmx(uint32_t t, uint32_t c) const {
        if (t >= live_end) {
            const uint32_t back = ((t - live_end) / cpt + 1) * cpt;
            if (back > t) return 0;
            t -= back;
        }
        return a.cm[(size_t)t * 256 + c];
    }
    // one wavefront per ctx, lane = subtile: 64 counts per trip, an inclusive scan across the lanes, 64 prefixes out (a
    // thread per ctx with sixteen loads a trip, or a thread per (subtile, ctx) that sums its own prefix: ~31 us a step)
    static size_t lds_bytes() { return 0; }
    template <class W>
    ORZ_D void operator()(W& w) const { run(w, w.block()); }
    template <class W>
    ORZ_D void run(W& w, uint32_t c) const {
        const uint32_t lane = w.lane();
        uint32_t carry = a.cp[(size_t)s0 * 256 + c];
        const uint32_t end = s1 + ext;
        for (uint32_t base = s0; base < end; base += 64) {
            const uint32_t s = base + lane;
            uint32_t v = s < end ? cmx(s, c) : 0;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t t = w.shfl(v, lane >= d ? lane - d : lane);
                if (lane >= d) v += t;
            }
            if (s < end) a.cp[(size_t)(s + 1) * 256 + c] = carry + v;
            carry += w.bcast(v, 63);
        }
    }
};

struct FastPrefixSerial {  // whole block, thread per ctx: sixteen independent loads in flight, then the chain of adds
    FastArgs a;
    uint32_t s0, s1;
    ORZ_HD void operator()(size_t c) const {
        if (c >= 256) return;
        uint32_t v = a.cp[(size_t)s0 * 256 + c];
        uint32_t s = s0;
        for (; s + 16 <= s1; s += 16) {
            uint32_t m[16];
            for (uint32_t k = 0; k < 16; k++) m[k] = a.cm[(size_t)(s + k) * 256 + c];
            for (uint32_t k = 0; k < 16; k++) { v += m[k]; a.cp[(size_t)(s + k + 1) * 256 + c] = v; }
        }
        for (; s < s1; s++) {
            v += a.cm[(size_t)s * 256 + c];
            a.cp[(size_t)(s + 1) * 256 + c] = v;
        }
    }
};

// ---- two kernels of a step in ONE launch (round 5) ---------------------------------------------------------------------------
// A step of the round loop is a chain of launches, most of them far too small to fill the GPU, and its end is not a chain: the
// ordinal prefix needs the path's item counts (PathMarkWave) but not the flips, the ring horizons need the prefix but not the
// retiring tile.  As branches of the graph on streams of their own these pairs depended on which hardware queues the runtime
// handed out; as ONE grid whose first blocks do one job and whose last blocks the other they always run side by
// side: FastFlip beside FastPrefix, then FastRetire beside FastHorizon.  Wave-shaped launches (64 threads a block).
// Round 6: a wavefront of the flip part takes 512 positions, not 64 -- a lane looks at eight positions with four 8-byte loads
// (FastFlip::flip_mask8), the positions that do have something to flip are listed in LDS (a prefix sum over the lanes' counts) and
// the wavefront then takes them a lane each: an eighth of the wavefronts, and no lane walks for eight positions in a row (a thread
// per eight positions WITHOUT the hand-over measured 45 -> 132 us in round 3).
constexpr uint32_t kFlipSpan = 512;
ORZ_HD size_t flip_blocks(uint32_t nflip) { return (nflip + kFlipSpan - 1) / kFlipSpan; }
struct FlipPrefixWave {
    FastFlip f;
    FastPrefix p;
    uint32_t nflip;  // positions FastFlip visits; its blocks come first
    static size_t lds_bytes() { return kFlipSpan * 2; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        const uint32_t fb = (uint32_t)flip_blocks(nflip);
        if (w.block() < fb) {
            uint16_t* q = (uint16_t*)w.lds();
            const uint32_t lane = w.lane(), base = w.block() * kFlipSpan, t0 = base + lane * 8;
            uint32_t m = t0 < nflip ? f.flip_mask8(t0) : 0;
            if (t0 + 8 > nflip && t0 < nflip) m &= (1u << (nflip - t0)) - 1;
            const uint32_t mine = (uint32_t)popc64(m);
            uint32_t v = mine;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t t = w.shfl(v, lane >= d ? lane - d : lane);
                if (lane >= d) v += t;
            }
            const uint32_t tot = w.bcast(v, 63);
            uint32_t off = v - mine;
            while (m) { const uint32_t k = (uint32_t)ctz64(m); m &= m - 1; q[off++] = (uint16_t)(lane * 8 + k); }
            w.sync();
            for (uint32_t e = lane; e < tot; e += 64) f((size_t)(base + q[e]));
        } else {
            p.run(w, w.block() - fb);
        }
    }
};
struct RetireHorizonWave {
    FastRetire r;
    FastHorizon h;
    uint32_t nret;  // positions of the retiring tile (0 = none this step); their blocks come first
    // (round 6 tried FlipPrefixWave's hand-over here too -- 512 positions a wavefront, the item starts among them a lane each: 2.1 ->
    // 2.8 ms a block.  A retiring item start is a chain of its own (galloping down the slots, bitmap words, the record), a third of the
    // positions are item starts, and an eighth of the wavefronts means an eighth of the chains in flight: a thread per position it stays.)
    static size_t lds_bytes() { return 0; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        const uint32_t rb = (nret + 63) / 64;
        if (w.block() < rb) {
            const uint32_t t = w.block() * 64 + w.lane();
            if (t < nret) r((size_t)t);
        } else {
            const size_t t = (size_t)(w.block() - rb) * 64 + w.lane();
            if (t < h.threads()) h(t);
        }
    }
};

// ---- after the rounds: sources, repairs, exact predictor ---------------------------------------------
// The repair passes run without the host in the loop: every kernel of a pass returns at once when an earlier pass found
// nothing to repair (`done`), FastPassEnd closes a pass on the device, and the host reads this block once per group of
// passes.
struct FastCtl {
    uint32_t chg;      // repairs of the running pass
    uint32_t done;     // a pass ended with none
    uint32_t total;    // repairs of all passes
    uint32_t passes;   // passes that did work
    uint32_t nmem;     // item starts when the last pass began (FastItemTotal)
    uint32_t acc;
    uint32_t lt;       // type of the item that ended at the block end (carried to the next block)
    uint32_t lastflips;  // item starts that still changed in their tile's LAST round (FastFlip): the rounds had not settled
    uint32_t nent;       // slots of the block's candidate lists -- read by the kernels of the round loop from HERE, not from their
                         // arguments: the loop is replayed as a graph whose arguments are those of the block it was captured on
    uint32_t ncut, nfix;  // entries of the running pass's cut list (FastSource -> FastRecutL) / WORD-fix list (FastWordCheckL -> FastWordApplyL)
    uint32_t nwx;         // ... and of the list of WORD items FastRecut made in this pass (the subtiles' lists were drawn up before it)
    uint32_t hot, hotacc; // item starts of the block's busiest ring context when the last pass began (the host picks the NEXT block's
                          // schedule from it, round 6) / its accumulator (FastItemTotal)
};
ORZ_HD bool passes_done(const FastCtl* ctl) { return ctl && ctl->done; }
struct FastCtlReset {
    FastCtl* ctl;
    ORZ_HD void operator()(size_t tid) const {
        if (tid) return;
        ctl->chg = 0; ctl->done = 0; ctl->total = 0; ctl->passes = 0; ctl->nmem = 0; ctl->acc = 0;  // (lastflips: reset before the rounds)
        ctl->ncut = 0; ctl->nfix = 0; ctl->nwx = 0; ctl->hot = 0; ctl->hotacc = 0;
    }
};
struct FastSetNent {
    FastCtl* ctl;
    uint32_t nent;
    ORZ_HD void operator()(size_t tid) const {
        if (!tid) ctl->nent = nent;
    }
};
struct FastPassEnd {
    FastCtl* ctl;
    ORZ_HD void operator()(size_t tid) const {
        if (tid || ctl->done) return;
        ctl->total += ctl->chg;
        ctl->passes++;
        if (ctl->chg == 0) ctl->done = 1;
        ctl->chg = 0;
        ctl->nmem = ctl->acc;
        ctl->acc = 0;
        ctl->hot = ctl->hotacc;
        ctl->hotacc = 0;
        ctl->ncut = 0; ctl->nfix = 0; ctl->nwx = 0;  // (the lists of the pass were consumed by FastRecutL / FastWordApplyL)
    }
};
// ---- the repair stage over lists (round 5) --------------------------------------------------------------------------------
// A pass used to be a dozen grids over all 2^24 positions of which a few thousand had work.  Now: per 4096-position subtile the
// items that a pass can touch are listed once per pass (RepairListWave: matches, WORD items -- u16 offsets, in position order,
// no atomics), the kernels that judge them run over those lists, what they decide goes to short lists with a counter (cuts,
// WORD fixes), and the positions whose bits the rewrite kernels change are noted in `tbits` so that the flips of the slot-order
// bitmaps visit those only (FastFlipSparse).  Same decisions, same bytes; the rule of the stage stands: a thread acts only on
// state written by an EARLIER launch.
constexpr uint32_t kSubMatches = kSub / kMinLen;  // a subtile holds at most 1024 matches (4 bytes each at least)
constexpr uint32_t kSubWords = kSub / 2;          // ... and 2048 WORD items
constexpr uint32_t kListThreads = 256;            // threads per subtile in the kernels that run over the lists
ORZ_D void touch(const FastArgs& a, uint32_t y) {  // y = window offset in [kPre, len]
    const uint32_t i = y - kPre;
    atom_or64(&a.tbits[i >> 6], 1ull << (i & 63));
}
struct FastFlipSparse {  // thread per word of tbits
    FastFlip f;          // (lo / hi unused; mark_hi = last_hi = 0, next_entry = ~0u)
    uint32_t nwords;     // words of tbits that can hold a bit: positions kPre .. len
    uint64_t* kdirty;
    const FastCtl* ctl = nullptr;
    ORZ_HD void operator()(size_t w) const {
        if (w >= nwords || passes_done(ctl)) return;
        uint64_t m = f.a.tbits[w];
        if (!m) return;
        f.a.tbits[w] = 0;
        while (m) {
            const uint32_t t = (uint32_t)ctz64(m);
            m &= m - 1;
            f.flip_one(kPre + (uint32_t)w * 64 + t, kdirty);
        }
    }
};
// One wavefront per subtile, lane = 64 positions: the subtile's item starts per ctx (what CountWave counts) and its matches
// and WORD items as lists of offsets inside the subtile, in position order (a prefix sum over the lanes' counts).
struct RepairListWave {
    FastArgs a;
    uint16_t *mlist, *wlist;  // [nsub][kSubMatches], [nsub][kSubWords]
    uint32_t *mcnt, *wcnt;    // [nsub]
    // later passes: only the matches FastSource has to look at again are listed -- those of a run that gained an item start in
    // the pass before (rdirty), those whose source lies near the end of the ring (edge) and those of a context that has grown
    // much since the first pass (cok): the tests FastSource makes itself, made here once per match while its bytes are at hand,
    // instead of by a thread per listed match (later passes of FastSourceL: 165 us -> the few that remain)
    const uint64_t* rdirty;   // nullptr: every match (the first pass)
    const uint8_t* edge;
    const uint32_t* cok;
    const FastCtl* ctl = nullptr;
    static size_t lds_bytes() { return 256 * 4; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        if (passes_done(ctl)) return;  // (the passes queued behind the one that found nothing walked the whole block all the same)
        uint32_t* cnt = (uint32_t*)w.lds();
        const uint32_t s = w.block(), lane = w.lane();
        for (uint32_t c = lane; c < 256; c += 64) cnt[c] = 0;
        w.sync();
        const uint32_t i0 = s * kSub + lane * 64;
        uint64_t mm = 0, wm = 0;  // the lane's matches / WORD items
        if (i0 < a.n) {
            uint64_t m = a.sbits[i0 / 64];
            const uint8_t* b = a.win + kPre + i0;
            while (m) {
                const uint32_t t = (uint32_t)ctz64(m);
                m &= m - 1;
                if (i0 + t >= a.n) break;
                atom_add32(&cnt[(uint32_t)(b[(int)t - 1] & 0x7f) | ((uint32_t)is_alnum(b[(int)t - 2]) << 7)], 1);
                const uint32_t ty = a.ty[i0 + t];
                if (ty == kTyMatch) mm |= 1ull << t;
                else if (ty == kTyWord) wm |= 1ull << t;
            }
            if (rdirty) {
                for (uint64_t q = mm; q;) {
                    const uint32_t t = (uint32_t)ctz64(q);
                    q &= q - 1;
                    const uint32_t p = kPre + i0 + t, c = hash1(a.win, p - 1), key = c * kHash + hash_entry(a.win + p);
                    if (!((rdirty[key >> 6] >> (key & 63)) & 1) && !edge[i0 + t] && cok[c]) mm &= ~(1ull << t);
                }
            }
        }
        // exclusive prefix of the two counts over the lanes (packed: matches in the low half)
        const uint32_t mine = (uint32_t)popc64(mm) | ((uint32_t)popc64(wm) << 16);
        uint32_t v = mine;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t t = w.shfl(v, lane >= d ? lane - d : lane);
            if (lane >= d) v += t;
        }
        const uint32_t tot = w.bcast(v, 63);
        uint32_t mo = (v - mine) & 0xffff, wo = (v - mine) >> 16;
        uint16_t* ml = mlist + (size_t)s * kSubMatches;
        uint16_t* wl = wlist + (size_t)s * kSubWords;
        while (mm) { const uint32_t t = (uint32_t)ctz64(mm); mm &= mm - 1; ml[mo++] = (uint16_t)(lane * 64 + t); }
        while (wm) { const uint32_t t = (uint32_t)ctz64(wm); wm &= wm - 1; wl[wo++] = (uint16_t)(lane * 64 + t); }
        if (lane == 0) { mcnt[s] = tot & 0xffff; wcnt[s] = tot >> 16; }
        w.sync();
        for (uint32_t c = lane; c < 256; c += 64) a.cm[(size_t)s * 256 + c] = cnt[c];
    }
};
struct FastItemTotal {  // item starts of the block from the per-ctx ordinals (thread per ctx)
    const uint32_t* cp;
    uint32_t nsub;
    FastCtl* ctl;
    ORZ_HD void operator()(size_t c) const {
        if (c >= 256 || ctl->done) return;
        const uint32_t mine = cp[(size_t)nsub * 256 + c] - cp[c];
        atom_add32(&ctl->acc, mine);
        atom_max32(&ctl->hotacc, mine);
    }
};
// Exact ring ordinals (Bucket.head arithmetic, src/matcher.rs:62-80) of the block's item starts without sorting them:
// ORD = cp[subtile][ctx] + rank among the subtile's earlier item starts of the same ctx.  One wavefront per subtile,
// lane = 64 positions: per-lane counts per ctx in LDS, an exclusive prefix down each ctx column, then every lane walks
// its item starts again.  Rows are padded to 257 entries against bank conflicts.
struct OrdWave {
    const uint8_t* win;
    const uint64_t* sbits;
    const uint32_t* cp;
    uint32_t n;
    uint32_t* ORD;
    const FastCtl* ctl;
    static constexpr uint32_t kRow = 257;
    static size_t lds_bytes() { return (size_t)64 * kRow * 2 + 64; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        if (ctl->done) return;
        uint16_t* cnt = (uint16_t*)w.lds();
        const uint32_t s = w.block(), lane = w.lane();
        uint16_t* mine = cnt + (size_t)lane * kRow;
        for (uint32_t c = 0; c < kRow; c++) mine[c] = 0;
        const uint32_t i0 = s * kSub + lane * 64;
        const uint64_t m0 = i0 < n ? sbits[i0 / 64] : 0;
        const uint8_t* b = win + kPre + i0;
        for (uint64_t m = m0; m;) {
            const uint32_t t = (uint32_t)ctz64(m);
            m &= m - 1;
            if (i0 + t < n) mine[(uint32_t)(b[(int)t - 1] & 0x7f) | ((uint32_t)is_alnum(b[(int)t - 2]) << 7)]++;
        }
        w.sync();
        for (uint32_t c = lane; c < 256; c += 64) {
            uint32_t run = 0;
            for (uint32_t r = 0; r < 64; r++) {
                const uint32_t v = cnt[(size_t)r * kRow + c];
                cnt[(size_t)r * kRow + c] = (uint16_t)run;
                run += v;
            }
        }
        w.sync();
        for (uint64_t m = m0; m;) {
            const uint32_t t = (uint32_t)ctz64(m);
            m &= m - 1;
            if (i0 + t >= n) break;
            const uint32_t c = (uint32_t)(b[(int)t - 1] & 0x7f) | ((uint32_t)is_alnum(b[(int)t - 2]) << 7);
            ORD[kPre + i0 + t] = cp[(size_t)s * 256 + c] + mine[c]++;
        }
    }
};

// The same ordinals with the wavefront's ballots instead of a 64 x 257 counter table (33 KB of LDS a wave: four waves a CU, and ~800
// LDS operations a lane): the subtile's item starts are listed in position order (prefix sum over the lanes' counts), then taken 64
// at a time -- a lane's rank among the batch's item starts of its own ctx comes from eight ballots (match-any on the ctx bits),
// the ordinals before the batch from a running count per ctx in LDS that the first lane of each ctx group moves on.  9 KB of LDS.
struct OrdWave2 {
    const uint8_t* win;
    const uint64_t* sbits;
    const uint32_t* cp;
    uint32_t n;
    uint32_t* ORD;
    const FastCtl* ctl;
    static size_t lds_bytes() { return 256 * 4 + kSub * 2; }
    template <class W>
    ORZ_D void operator()(W& w) const {
        if (ctl->done) return;
        uint32_t* cnt = (uint32_t*)w.lds();
        uint16_t* pos = (uint16_t*)(cnt + 256);
        const uint32_t s = w.block(), lane = w.lane();
        for (uint32_t c = lane; c < 256; c += 64) cnt[c] = cp[(size_t)s * 256 + c];
        const uint32_t i0 = s * kSub + lane * 64;
        uint64_t m = i0 < n ? sbits[i0 / 64] : 0;
        if (i0 < n && n - i0 < 64) m &= (1ull << (n - i0)) - 1;
        const uint32_t mine = (uint32_t)popc64(m);
        uint32_t v = mine;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t t = w.shfl(v, lane >= d ? lane - d : lane);
            if (lane >= d) v += t;
        }
        const uint32_t tot = w.bcast(v, 63);
        uint32_t off = v - mine;
        while (m) { const uint32_t t = (uint32_t)ctz64(m); m &= m - 1; pos[off++] = (uint16_t)(lane * 64 + t); }
        w.sync();
        for (uint32_t base = 0; base < tot; base += 64) {
            const uint32_t k = base + lane;
            const bool valid = k < tot;
            const uint32_t x = kPre + s * kSub + (valid ? pos[k] : 0);
            const uint32_t c = valid ? hash1(win, x - 1) : 0;
            uint64_t peers = w.ballot(valid);
            for (uint32_t b = 0; b < 8; b++) {
                const uint64_t bb = w.ballot(valid && ((c >> b) & 1));
                peers &= ((c >> b) & 1) ? bb : ~bb;
            }
            const uint32_t rank = (uint32_t)popc64(peers & ((1ull << lane) - 1));
            uint32_t before = 0;
            if (valid) { before = cnt[c]; ORD[x] = before + rank; }
            w.sync();
            if (valid && rank == 0) cnt[c] = before + (uint32_t)popc64(peers);
            w.sync();
        }
    }
};

struct FastSource {
    FastArgs a;
    uint32_t* SRC;
    uint32_t* cutend;  // [n] old end of an item that lost part of its length in this pass (0 = none)
    // Later passes (rdirty != nullptr) walk again only where the answer can have changed: the newest covering item start of
    // a run is still the newest unless the run gained an item start in the previous pass (rdirty, set by FastRecut /
    // FastWordCheck -- a new item's own run is marked, so new items are walked too), and it is still a ring member unless
    // the ordinals between it and the item grew past the ring (checked here with the fresh ordinals).
    // ... and the ring check costs two dependent scattered loads per match, so a pass notes for every source it assigns
    // whether it lies within kEdgeMargin item starts of the ring's end (`edge`); a later pass looks at the ordinals only
    // for those, or when the context has gained more than kEdgeMargin item starts since the first pass (`cok`, FastCtxOk).
    const uint64_t* rdirty;
    uint32_t cap;  // item starts examined beyond the tabulated window before the search gives up (the item is cut then)
    const FastCtl* ctl;
    uint8_t* edge;         // [n] the match's source is near the end of the ring
    const uint32_t* cok;   // [256] the context has gained at most kEdgeMargin item starts since the first pass
    uint32_t* cutlist = nullptr;  // the items cut in this pass (FastRecutL; nullptr: FastRecut looks at every position)
    uint32_t* ncut = nullptr;
    static constexpr uint32_t kEdgeMargin = 1024;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || ctl->done || !((a.sbits[i / 64] >> (i & 63)) & 1) || a.ty[i] != kTyMatch) return;
        const uint32_t p = kPre + (uint32_t)i, L = a.nl[i], K = a.K;
        if (rdirty) {
            const uint32_t c = hash1(a.win, p - 1), key = c * kHash + hash_entry(a.win + p);
            if (!((rdirty[key >> 6] >> (key & 63)) & 1)) {
                if (!edge[i] && cok[c]) return;
                if (a.ORD[p] - 1 - a.ORD[SRC[p]] <= kRing - 1) return;
            }
        }
        const uint32_t op = a.ORD[p];
        const uint32_t j = a.idx[p], r = fast_min(K, a.rlen[i]);
        const uint8_t* row = a.rows + (size_t)i * K;
        uint32_t best = 0, bsrc = 0, seen = 0, found = 0;
        bool stop = false;
        for (uint32_t m = 0; m * 64 < r && !stop && !found; m++) {  // (no depth limit here: any ring member may serve)
            uint64_t mask = bits_at(a.vbits, (int64_t)j - 64 * (int64_t)(m + 1));
            const uint32_t span = r - m * 64;
            if (span < 64) mask &= ~0ull << (64 - span);
            while (mask) {
                const uint32_t t = 63 - (uint32_t)clz64(mask);
                mask &= ~(1ull << t);
                const uint32_t k = m * 64 + 63 - t;
                const uint32_t l = row[k];
                if (l >= kMinLen && (l >= L || l > best)) {
                    const uint32_t q = a.epos[j - 1 - k];
                    if (op - 1 - a.ORD[q] > kRing - 1) { stop = true; break; }  // left the ring: so did everything older
                    if (l >= L) { found = q; break; }
                    best = l; bsrc = q;
                }
                seen++;
            }
        }
        if (!found && !stop && a.rlen[i] > K && a.far) {
            const uint32_t rs = a.rlen[i] < 255 ? j - a.rlen[i] : a.runstart[bucket_key(a.win, p)];
            const uint32_t top = j - K;
            const uint32_t lo2 = top - rs > a.far ? top - a.far : rs;
            const uint64_t a0 = a.stext[2 * (size_t)j], a1 = a.stext[2 * (size_t)j + 1];
            uint32_t left = cap;
            far_walk(a, lo2, top, [&](uint32_t sl) -> bool {
                uint32_t q;
                const uint32_t l = far_lcp(a, p, a0, a1, sl, &q);
                if (l >= kMinLen && (l >= L || l > best)) {
                    if (op - 1 - a.ORD[q] > kRing - 1) { stop = true; return false; }
                    if (l >= L) { found = q; return false; }
                    best = l; bsrc = q;
                }
                return --left != 0;
            });
        }
        if (found) { SRC[p] = found; edge[i] = op - 1 - a.ORD[found] > kRing - 1 - kEdgeMargin; return; }
        atom_add32(a.nchg, 1);
        cutend[i] = p + L;
        if (cutlist) cutlist[atom_fetch_add32(ncut, 1)] = (uint32_t)i;
        if (best >= kMinLen) { a.nl[i] = (uint8_t)best; SRC[p] = bsrc; edge[i] = op - 1 - a.ORD[bsrc] > kRing - 1 - kEdgeMargin; }
        else { a.ty[i] = kTyLit; a.nl[i] = 1; }
    }
};
struct FastCtxOk {  // thread per ctx: item starts of the context now against the first pass's (FastSource's `cok`)
    const uint32_t* cp;
    uint32_t nsub;
    uint32_t* tot0;  // [256] totals at the first pass
    uint32_t* cok;   // [256]
    int first;
    const FastCtl* ctl;
    ORZ_HD void operator()(size_t c) const {
        if (c >= 256 || ctl->done) return;
        const uint32_t t = cp[(size_t)nsub * 256 + c];
        if (first) { tot0[c] = t; cok[c] = 1; }
        else cok[c] = t - tot0[c] <= FastSource::kEdgeMargin;
    }
};
struct FastRecut {  // the rest of a shortened item's span, from the last round's decisions, cut at the old end
    FastArgs a;
    uint32_t* cutend;
    uint64_t* rdirty;  // runs that gain an item start here (read by the next pass's FastSource)
    uint32_t* cgrow = nullptr;   // list form: [256] item starts added per context since the first pass (FastCokGrow)
    uint32_t* wextra = nullptr;  // list form: the WORD items made here (this pass's FastWordCheckL judges them: the subtiles' lists
    uint32_t* nwx = nullptr;     // were drawn up before); the positions rewritten are noted in a.tbits for FastFlipSparse
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || !cutend[i]) return;  // (nothing is cut once the passes are done)
        const uint32_t end = cutend[i];
        cutend[i] = 0;
        uint32_t x = kPre + (uint32_t)i + a.nl[i];
        a.pt[x - kPre] = a.ty[i];
        if (wextra) touch(a, x);
        while (x < end) {
            const uint32_t xi = x - kPre;
            uint32_t t = a.ty[xi], L = a.nl[xi];
            if (L == 0) { t = kTyLit; L = 1; }
            if (x + L > end) {
                if (t == kTyMatch && end - x >= kMinLen) L = end - x;
                else { t = kTyLit; L = 1; }
            }
            a.ty[xi] = (uint8_t)t; a.nl[xi] = (uint8_t)L;
            atom_or64(&a.sbits[xi / 64], 1ull << (xi & 63));
            mark_run(a.win, rdirty, x);
            if (wextra && t == kTyWord) wextra[atom_fetch_add32(nwx, 1)] = xi;
            if (cgrow) atom_add32(&cgrow[hash1(a.win, x - 1)], 1);
            x += L;
            a.pt[x - kPre] = (uint8_t)t;
            if (wextra) touch(a, x);  // (the item start at x - L was touched as the end of the item before it)
        }
    }
};
struct FastRecutL {  // the same over the pass's cut list
    FastRecut f;
    const uint32_t* cutlist;
    const FastCtl* ctl;
    uint32_t nthreads;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t n = ctl->ncut;
        for (uint32_t k = (uint32_t)tid; k < n; k += nthreads) f(cutlist[k]);
    }
};
struct FastSourceL {  // FastSource over the subtiles' match lists: `per` threads a subtile (the first pass, every match: one thread a
    FastSource f;     // list slot -- a thread that takes several matches walks for them one after the other; later passes: a few)
    const uint16_t* mlist;
    const uint32_t* mcnt;
    uint32_t nsub, per;
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t s = (uint32_t)(tid / per);
        if (s >= nsub || f.ctl->done) return;
        const uint32_t c = mcnt[s];
        for (uint32_t k = (uint32_t)(tid % per); k < c; k += per) f((size_t)s * kSub + mlist[(size_t)s * kSubMatches + k]);
    }
};
struct KbitVals {  // v[s] = s + 1 where the word-update bit of slot s is set, else 0 (for the running maximum)
    const uint64_t* kbits;
    uint32_t nk;
    uint32_t* v;
    ORZ_HD void operator()(size_t s) const {
        if (s < nk) v[s] = ((kbits[s >> 6] >> (s & 63)) & 1) ? (uint32_t)s + 1 : 0;
    }
};
// exact words[] answer for an item starting at p (src/lz.rs:132-133), from the running maximum `laste`
ORZ_D uint32_t fast_word_at(const FastArgs& a, const uint32_t* laste, uint32_t p) {
    const uint32_t key2 = hash2(a.win, p - 1);
    const uint32_t kj = a.kidx[p], klo = a.krun[key2];
    uint32_t below = kj;  // slots [klo, below) are the candidates
    if (kj > klo && hash2(a.win, p - 2) == key2) below = kj - 1;  // u = p-1 does not count yet
    const uint32_t le = below > klo ? laste[below - 1] : 0;
    if (le > klo) return a.kw[le - 1];
    return (uint32_t)a.wsnap[key2 * 2] | ((uint32_t)a.wsnap[key2 * 2 + 1] << 8);
}
// A WORD item whose prediction the exact state does not make becomes two literals -- in TWO launches.  FastWordCheck only
// NOTES the verdict (cutend[i] = kWordFix; the array is all zero between FastRecut and the next FastSource), FastWordApply
// rewrites the item.  As one kernel this was the round-3 defect (1 of 10^2..10^3 members of 64 MiB
// undecodable, on the GPU only): the rewrite makes i + 1 an item start, and the thread of i + 1 -- in another wavefront when
// i is a wavefront's last lane, running at the same time -- could see that new bit in `sbits` but still the ROUND's decision
// in ty[i + 1] (the plain store had not reached it).  Was that stale decision WORD with a wrong prediction, it "repaired" the
// item that is not one: ty[i + 2] = literal -- which cut a match at i + 2 down to one byte with nothing re-parsing its span, a
// hole in the path that no decoder follows.  The host emulation runs a launch's threads one after another and never saw it.
// Rule for every kernel of the repair stage: a thread acts only on state that was written by an EARLIER launch.
constexpr uint32_t kWordFix = ~0u;
struct FastWordCheck {
    FastArgs a;
    const uint32_t* laste;
    uint32_t* fix;  // [n] = cutend: kWordFix where a WORD item has to go
    const FastCtl* ctl;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || ctl->done || !((a.sbits[i / 64] >> (i & 63)) & 1) || a.ty[i] != kTyWord) return;
        const uint32_t p = kPre + (uint32_t)i;
        const uint32_t w = fast_word_at(a, laste, p);
        if (w == ((uint32_t)a.win[p] | ((uint32_t)a.win[p + 1] << 8))) return;
        fix[i] = kWordFix;
    }
};
struct FastWordApply {  // thread per position: acts on the flags of the launch before, writes nothing another thread of this launch reads
    FastArgs a;
    uint32_t* fix;
    uint64_t* rdirty;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || fix[i] != kWordFix) return;
        fix[i] = 0;
        const uint32_t p = kPre + (uint32_t)i;
        atom_add32(a.nchg, 1);
        a.ty[i] = kTyLit; a.nl[i] = 1;
        a.ty[i + 1] = kTyLit; a.nl[i + 1] = 1;
        atom_or64(&a.sbits[(i + 1) / 64], 1ull << ((i + 1) & 63));
        mark_run(a.win, rdirty, p + 1);
        a.pt[i + 1] = kTyLit; a.pt[i + 2] = kTyLit;
    }
};
// The newest word update among the slots [klo, below) of the word list WITHOUT the running maximum: the highest set bit of
// kbits there, through the summary level (returns slot + 1, 0 = none).  The later repair passes ask for a handful of WORD
// items only (those of keys whose update bits changed); a scan over all 2^24 slots per pass was 0.1 ms of each.
ORZ_D uint32_t kbits_prev(const FastArgs& a, uint32_t klo, uint32_t below) {
    if (below <= klo) return 0;
    const uint32_t wlo = klo >> 6;
    uint32_t wi = (below - 1) >> 6;
    uint64_t m = a.kbits[wi];
    if (below & 63) m &= (1ull << (below & 63)) - 1;
    for (;;) {
        if (wi == wlo) m &= ~0ull << (klo & 63);
        if (m) return wi * 64 + 63 - (uint32_t)clz64(m) + 1;
        if (wi == wlo) return 0;
        // the next non-empty word below wi, by the summary (bit per word; a stale set bit only costs the load of an empty word)
        uint32_t nw = wi - 1;
        uint64_t sm = a.k1[nw >> 6];
        if ((nw & 63) != 63) sm &= (2ull << (nw & 63)) - 1;
        while (!sm) {
            if ((nw >> 6) == 0 || (nw >> 6) <= (wlo >> 6)) return 0;
            nw = ((nw >> 6) << 6) - 1;
            sm = a.k1[nw >> 6];
        }
        wi = (nw & ~63u) + 63 - (uint32_t)clz64(sm);
        if (wi < wlo) return 0;
        m = a.kbits[wi];
    }
}
ORZ_D uint32_t fast_word_search(const FastArgs& a, uint32_t p) {  // = fast_word_at, from the bitmap itself
    const uint32_t key2 = hash2(a.win, p - 1);
    const uint32_t kj = a.kidx[p], klo = a.krun[key2];
    uint32_t below = kj;
    if (kj > klo && hash2(a.win, p - 2) == key2) below = kj - 1;
    const uint32_t le = kbits_prev(a, klo, below);
    if (le > klo) return a.kw[le - 1];
    return (uint32_t)a.wsnap[key2 * 2] | ((uint32_t)a.wsnap[key2 * 2 + 1] << 8);
}
struct FastWordCheckL {  // FastWordCheck over the subtiles' WORD lists; the verdicts go to a list
    FastArgs a;
    const uint32_t* laste;   // the first pass: the running maximum, every WORD item; later passes (nullptr): only the items of
    const uint64_t* kdirty;  // keys whose update bits changed since they were judged (FastFlipSparse), by search -- and the
    const uint32_t* wextra;  // WORD items this pass's FastRecut made, whatever their keys
    uint32_t* fixlist;
    FastCtl* ctl;
    const uint16_t* wlist;
    const uint32_t* wcnt;
    uint32_t nsub;
    ORZ_D void check(uint32_t i, bool fresh) const {
        if (!((a.sbits[i / 64] >> (i & 63)) & 1) || a.ty[i] != kTyWord) return;
        const uint32_t p = kPre + i;
        uint32_t w;
        if (laste) w = fast_word_at(a, laste, p);
        else {
            const uint32_t key2 = hash2(a.win, p - 1);
            if (!fresh && !((kdirty[key2 >> 6] >> (key2 & 63)) & 1)) return;
            w = fast_word_search(a, p);
        }
        if (w == ((uint32_t)a.win[p] | ((uint32_t)a.win[p + 1] << 8))) return;
        fixlist[atom_fetch_add32(&ctl->nfix, 1)] = i;
    }
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t s = (uint32_t)(tid / kListThreads);
        if (s >= nsub || ctl->done) return;
        const uint32_t c = wcnt[s];
        for (uint32_t k = (uint32_t)(tid % kListThreads); k < c; k += kListThreads) check(s * kSub + wlist[(size_t)s * kSubWords + k], false);
        const uint32_t nx = ctl->nwx, nth = nsub * kListThreads;
        for (uint32_t k = (uint32_t)tid; k < nx; k += nth) check(wextra[k], true);
    }
};
struct FastCokGrow {  // thread per ctx: has the context gained at most kEdgeMargin item starts since the first pass? (= FastCtxOk, from
    const uint32_t* cgrow;  // the counters FastRecut / FastWordApplyL keep instead of the pass's ordinals)
    uint32_t* cok;
    // round 6: the launch also clears the run flags this pass will set (`rd_out`: nwords words, nullptr = cleared elsewhere) --
    // it was a fill dispatch of its own in every pass; nothing reads them between the pass before's FastSourceL and this
    // pass's FastRecutL.  Returns at once when the passes are done.
    uint64_t* rd_out = nullptr;
    uint32_t nwords = 0;
    const FastCtl* ctl = nullptr;
    ORZ_HD size_t threads() const { return nwords > 256 ? nwords : 256; }
    ORZ_HD void operator()(size_t c) const {
        if (ctl && ctl->done) return;
        if (c < 256) cok[c] = cgrow[c] <= FastSource::kEdgeMargin;
        if (rd_out && c < nwords) rd_out[c] = 0;
    }
};
struct FastWordApplyL {  // FastWordApply over the fix list (a launch of its own: see FastWordCheck)
    FastArgs a;
    const uint32_t* fixlist;
    uint64_t* rdirty;
    uint32_t* cgrow;
    const FastCtl* ctl;
    uint32_t nthreads;
    uint64_t* kdirty = nullptr;  // [512] round 6: cleared here (FastWordCheckL, the launch before, was its last reader of the pass) instead
                                 // of by a fill dispatch of its own
    ORZ_HD void operator()(size_t tid) const {
        if (kdirty && tid < 512) kdirty[tid] = 0;
        const uint32_t n = ctl->nfix;
        for (uint32_t k = (uint32_t)tid; k < n; k += nthreads) {
            const uint32_t i = fixlist[k], p = kPre + i;
            atom_add32(a.nchg, 1);
            a.ty[i] = kTyLit; a.nl[i] = 1;
            a.ty[i + 1] = kTyLit; a.nl[i + 1] = 1;
            atom_or64(&a.sbits[(i + 1) / 64], 1ull << ((i + 1) & 63));
            mark_run(a.win, rdirty, p + 1);
            a.pt[i + 1] = kTyLit; a.pt[i + 2] = kTyLit;
            touch(a, p + 1); touch(a, p + 2);
            atom_add32(&cgrow[hash1(a.win, p)], 1);  // (the new item start at p + 1)
        }
    }
};
#if defined(ORZ_RACE_SELFTEST)
// (tests/race only) the one-kernel form of round 3, kept to show that the race check reports it: the thread of i + 1 reads
// sbits / ty[i + 1] while the thread of i -- another wavefront when i is a wavefront's last lane -- rewrites them
struct FastWordCheckRacy {
    FastArgs a;
    const uint32_t* laste;
    uint64_t* rdirty;
    const FastCtl* ctl;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || ctl->done || !((a.sbits[i / 64] >> (i & 63)) & 1) || a.ty[i] != kTyWord) return;
        const uint32_t p = kPre + (uint32_t)i;
        const uint32_t w = fast_word_at(a, laste, p);
        if (w == ((uint32_t)a.win[p] | ((uint32_t)a.win[p + 1] << 8))) return;
        atom_add32(a.nchg, 1);
        a.ty[i] = kTyLit; a.nl[i] = 1;
        a.ty[i + 1] = kTyLit; a.nl[i + 1] = 1;
        atom_or64(&a.sbits[(i + 1) / 64], 1ull << ((i + 1) & 63));
        mark_run(a.win, rdirty, p + 1);
        a.pt[i + 1] = kTyLit; a.pt[i + 2] = kTyLit;
    }
};
#endif
// Diagnostics (ORZ_FAST_VERIFY): every match of the frozen parse against first principles -- the source is an item start of
// the same context, lies inside the ring by the exact ordinals, and its bytes equal the item's -- independent of the
// tables the source assignment used.  err[0..4] = matches checked, source not an item start, other context, outside the
// ring, bytes differ; err[5] = position of the first offender.
struct FastVerify {
    FastArgs a;
    const uint32_t* SRC;
    const uint8_t* S;  // history item starts
    unsigned long long* err;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n || !((a.sbits[i / 64] >> (i & 63)) & 1) || a.ty[i] != kTyMatch) return;
        const uint32_t p = kPre + (uint32_t)i, L = a.nl[i], q = SRC[p];
        atom_add64(&err[0], 1);
        bool bad = false;
        const bool member = q < p && (q >= kPre ? ((a.sbits[(q - kPre) / 64] >> ((q - kPre) & 63)) & 1) != 0 : (q >= 1 && S[q] != 0));
        if (!member) { atom_add64(&err[1], 1); bad = true; }
        else {
            if (hash1(a.win, q - 1) != hash1(a.win, p - 1)) { atom_add64(&err[2], 1); bad = true; }
            if (a.ORD[p] - 1 - a.ORD[q] > kRing - 1) { atom_add64(&err[3], 1); bad = true; }
            bool same = L >= kMinLen && L <= kMaxLen;
            for (uint32_t k = 0; k < L && same; k++) same = a.win[q + k] == a.win[p + k];
            if (!same) { atom_add64(&err[4], 1); bad = true; }
        }
        if (bad) err[5] = p;
    }
};
struct FastCommit {  // per-position arrays the post stage reads (orz_stream.h)
    FastArgs a;
    const uint32_t* laste;
    const uint32_t* lt0p;  // type of the item that ended at the block start
    uint8_t *S, *TY, *ML, *W0;
    ORZ_HD void operator()(size_t i) const {
        if (i >= a.n) return;
        const uint32_t p = kPre + (uint32_t)i;
        if (!((a.sbits[i / 64] >> (i & 63)) & 1)) { S[p] = 0; ML[p] = 0; return; }
        const uint32_t t = a.ty[i], prev = i ? a.pt[i] : *lt0p;
        S[p] = 1;
        TY[p] = (uint8_t)(t | ((prev == kTyLit) << 2));
        ML[p] = t == kTyMatch ? a.nl[i] : 0;
        W0[p] = (uint8_t)(fast_word_at(a, laste, p) & 0xff);
    }
};
struct FastWordsCarry {  // words[] for the next block: per hash2 key the last update of this block
    FastArgs a;
    const uint32_t *laste, *krunend;
    uint8_t* wsnap;
    ORZ_HD void operator()(size_t key) const {
        if (key >= 32768) return;
        const uint32_t lo = a.krun[key], hi = krunend[key];
        if (hi <= lo) return;
        const uint32_t le = laste[hi - 1];
        if (le > lo) { const uint32_t w = a.kw[le - 1]; wsnap[key * 2] = (uint8_t)w; wsnap[key * 2 + 1] = (uint8_t)(w >> 8); }
    }
};
struct FastCtxCarry {
    const uint32_t* cp;
    uint32_t nsub;
    uint32_t* ctxcount;
    ORZ_HD void operator()(size_t c) const {
        if (c < 256) ctxcount[c] = cp[(size_t)nsub * 256 + c];
    }
};
struct FastCpInit {
    const uint32_t* ctxcount;
    uint32_t *cp, *tentry;
    ORZ_HD void operator()(size_t c) const {
        if (c < 256) cp[c] = ctxcount[c];
        if (c == 0) tentry[0] = kPre;  // the path enters the first tile at the block start
    }
};
struct FastLtCarry {  // the type of the item that ends at the block end goes to the next block (runs after FastCommit)
    const uint8_t* pt;
    uint32_t n;
    FastCtl* ctl;
    ORZ_HD void operator()(size_t tid) const {
        if (!tid) ctl->lt = pt[n];
    }
};

}  // namespace orz
