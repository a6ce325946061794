// orz_kernels.h -- kernel bodies of the MI355X ROLZ encoder (one functor per kernel).
//
// Each functor's operator()(tid) is the per-thread body; the HIP backend launches it as a
// __global__ grid on gfx950, the emulation backend (tests only) runs it in a host loop.
//
// What replaces what (reference = /root/reference, Rust):
//   (the parse itself -- find_match / has_lazy_match / ring updates -- lives in orz_parse.h)
//   LenMin*       Bucket::update's match_len_min rule, src/matcher.rs:65-71
//   ItemSyms      symbol construction src/lz.rs:148,173-189,216-230
//   SymRank       SymRankCoder src/symrank.rs:38-97 driven by src/lz.rs:274-305
//   Census        first-chunk census src/lz.rs:238-265
//   Hist          huff_weights src/lz.rs:272-305
//   HuffBuild     HuffmanTable::new_from_sym_weights + HuffmanEncoding src/huffman.rs:27-141
//   Header / Pack Encoder src/coder.rs:27-89 + src/lz.rs:253-256,268-269,311-342
#pragma once
#include "orz_common.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define ORZ_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define ORZ_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define ORZ_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define ORZ_ATOMIC_OR(p, v) atomicOr((p), (v))
#else  // host emulation: real (relaxed) atomics too -- the race check of tests/race runs a launch's threads on several host threads
template <class T, class U> inline T orz_fetch_add(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> inline T orz_fetch_min(T* p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T, class U> inline T orz_fetch_max(T* p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v > o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T, class U> inline T orz_fetch_or(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
#define ORZ_ATOMIC_ADD(p, v) orz_fetch_add((p), (v))
#define ORZ_ATOMIC_MIN(p, v) orz_fetch_min((p), (v))
#define ORZ_ATOMIC_MAX(p, v) orz_fetch_max((p), (v))
#define ORZ_ATOMIC_OR(p, v) orz_fetch_or((p), (v))
#endif

namespace orz {

// ---------------------------------------------------------------------------------------------
// Post-parse: items
// match_len_min of the source at the time of each reference (src/matcher.rs:65-71): the ring
// node's value is min(127, 1 + max len of earlier references), 0 if none.
struct LenMinKeys {  // key = src << 25 | pos for match items, ~0 for the others
    const uint32_t* ipos;
    const uint8_t* TY;
    const uint32_t* SRC;
    uint32_t nitems;
    uint64_t* keys;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nitems) return;
        uint32_t p = ipos[tid];
        keys[tid] = (TY[p] & 3) == kTyMatch ? (((uint64_t)SRC[p] << kPosBits) | p) : ~0ull;
    }
};
struct LenMinEval {  // thread per sorted reference: exclusive prefix max over its source's run
    const uint64_t* keys;
    uint32_t nitems;
    const uint8_t* ML;
    const uint8_t* LENMIN;  // carried value per position (history sources), 0 for new ones
    uint8_t* LMV;           // out: len_min seen by the reference at position p
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nitems) return;
        uint64_t k = keys[tid];
        if (k == ~0ull) return;
        uint32_t q = (uint32_t)(k >> kPosBits), p = (uint32_t)(k & kPosMask);
        uint32_t v = LENMIN[q];
        for (size_t j = tid; j > 0;) {
            j--;
            uint64_t kj = keys[j];
            if ((uint32_t)(kj >> kPosBits) != q) break;
            uint32_t l = ML[(uint32_t)(kj & kPosMask)];
            uint32_t w = l + 1 < 127 ? l + 1 : 127;
            if (w > v) v = w;
        }
        LMV[p] = (uint8_t)v;
    }
};
struct LenMinCommit {  // last reference of each run folds the whole run into LENMIN[src]
    const uint64_t* keys;
    uint32_t nitems;
    const uint8_t* ML;
    const uint8_t* LMV;
    uint8_t* LENMIN;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nitems) return;
        uint64_t k = keys[tid];
        if (k == ~0ull) return;
        uint32_t q = (uint32_t)(k >> kPosBits), p = (uint32_t)(k & kPosMask);
        if (tid + 1 < nitems && (uint32_t)(keys[tid + 1] >> kPosBits) == q && keys[tid + 1] != ~0ull) return;
        uint32_t v = LMV[p], l = ML[p];
        uint32_t w = l + 1 < 127 ? l + 1 : 127;
        LENMIN[q] = (uint8_t)(w > v ? w : v);
    }
};

// raw symbols, contexts and match side info per item (src/lz.rs:134-136,148,173-189,216-230)
struct ItemSyms {
    const uint8_t* win;
    const uint32_t* ipos;
    uint32_t nitems;
    const uint8_t* TY;
    const uint8_t* ML;
    const uint8_t* W0;
    const uint8_t* LMV;
    const uint32_t* SRC;
    const uint32_t* ORD;
    uint16_t* isym;   // raw symbol
    uint16_t* ictx;   // symrank context (9 bit)
    uint8_t* iunl;    // unlikely symbol
    uint8_t* ienc;    // encoded_match_len
    uint16_t* irob;   // robits | robitlen << 12
    uint8_t* ial;     // after_literal | is_match << 1
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nitems) return;
        uint32_t p = ipos[tid];
        uint32_t ty = TY[p] & 3, al = (TY[p] >> 2) & 1;
        ictx[tid] = (uint16_t)(hash1(win, p - 1) | (al << 8));
        iunl[tid] = W0[p];
        ial[tid] = (uint8_t)(al | ((ty == kTyMatch) << 1));
        ienc[tid] = 0;
        irob[tid] = 0;
        if (ty == kTyMatch) {
            uint32_t q = SRC[p], L = ML[p];
            uint32_t ro = ORD[p] - 1 - ORD[q];
            uint32_t m = LMV[p] > kMinLen ? LMV[p] : kMinLen;
            uint32_t e = ML[q] > kMinLen ? ML[q] : kMinLen;
            uint32_t enc = L > e ? L - m : (L < e ? L - m + 1 : 0);
            // (a ring distance beyond the ring or a length below len_min cannot be coded: only a defective parse gets here, the
            // validity gate -- orz_verify.h -- reports it and fails the encode; the clamps keep the table indices of the later
            // kernels inside their tables until then)
            if (ro > kRing - 1) ro = kRing - 1;
            if (enc > kLenSyms - 1) enc = kLenSyms - 1;
            uint32_t roid, bl, bits;
            roid_encode(ro, &roid, &bl, &bits);
            uint32_t lenid = enc < 5 ? enc : 5;
            isym[tid] = (uint16_t)(256 + roid * 6 + lenid);
            ienc[tid] = (uint8_t)enc;
            irob[tid] = (uint16_t)(bits | (bl << 12));
        } else if (ty == kTyWord) {
            isym[tid] = kWordSym;
        } else {
            isym[tid] = win[p];
        }
    }
};

// first chunk of the stream: symbol census (src/lz.rs:240-244)
struct CensusCount {
    const uint16_t* isym;
    uint32_t n;
    uint32_t* counts;  // [389]
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) ORZ_ATOMIC_ADD(&counts[isym[tid]], 1u);
    }
};
// symrank table state: per context {value[389], index[389], cnt, sum}
constexpr uint32_t kSrWords = kSyms * 2 + 4;
// The one property of a block's ranks that can be checked without the tables (src/symrank.rs:38-47): rank 388 is written
// exactly where the symbol IS the excluded one, and no rank exceeds it.  gsym = symbol | excluded symbol << 16, in the
// order the ranks were produced in.
struct SymCheck {
    const uint32_t* gsym;
    const uint16_t* grank;
    uint32_t nitems;
    uint32_t* flag;           // += violations
    const uint32_t* only_if;  // (second check: nothing to do unless the first one found something)
    static constexpr uint32_t kThreads = 65536;  // a small grid that strides over the items: these launches sit on the
                                                 // ranking chain and must not queue behind the parse kernels' dispatch
    ORZ_HD void operator()(size_t t) const {
        if (t >= kThreads || (only_if && *only_if == 0)) return;
        uint32_t bad = 0;
        for (size_t k = t; k < nitems; k += kThreads) {
            const uint32_t g = gsym[k], r = grank[k];
            bad += r > kSyms - 1 || ((r == kSyms - 1) != ((g & 0xffff) == (g >> 16)));
        }
        if (bad) ORZ_ATOMIC_ADD(flag, bad);
    }
};
struct SymGuardBegin {  // the tables as they are before a block's ranking -> backup; flags cleared (a kernel, not a copy engine)
    const uint64_t* state;
    uint64_t* backup;
    uint32_t nwords;  // 8-byte words
    uint32_t* flags;
    ORZ_HD void operator()(size_t t) const {
        if (t < nwords) backup[t] = state[t];
        if (t == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; }
    }
};
struct SymCompare {  // (diagnostics, ORZ_SYMRANK_VERIFY) the ranks of two runs of a block's ranking side by side
    const uint16_t* a;
    const uint16_t* b;
    uint32_t nitems;
    uint32_t* flag;  // += differences
    static constexpr uint32_t kThreads = 65536;
    ORZ_HD void operator()(size_t t) const {
        if (t >= kThreads) return;
        uint32_t bad = 0;
        for (size_t k = t; k < nitems; k += kThreads) bad += a[k] != b[k];
        if (bad) ORZ_ATOMIC_ADD(flag, bad);
    }
};
struct SymKeep {  // (diagnostics) the first run's ranks, put aside
    const uint16_t* src;
    uint16_t* dst;
    uint32_t nitems;
    ORZ_HD void operator()(size_t t) const {
        for (size_t k = t; k < nitems; k += SymCompare::kThreads) dst[k] = src[k];
    }
};
struct SymInject {  // (tests) the failure the guard exists for: item k's rank reads "excluded symbol" although it is not
    const uint32_t* gsym;
    uint16_t* grank;
    uint32_t k;
    ORZ_HD void operator()(size_t tid) const {
        if (tid == 0 && (gsym[k] & 0xffff) != (gsym[k] >> 16)) grank[k] = (uint16_t)(kSyms - 1);
    }
};
// Stable order by descending max(count, 1) (src/lz.rs:247-250): thread per symbol, its place is the
// number of symbols that sort before it.
struct CensusOrder {
    const uint32_t* counts;
    uint16_t* order;     // [389] out: initial rank order
    uint32_t* ncounted;  // out: symbols with count > 1 (src/lz.rs:253)
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= kSyms) return;
        const uint32_t key = counts[tid] > 1 ? counts[tid] : 1;
        uint32_t place = 0, k = 0;
        for (uint32_t t = 0; t < kSyms; t++) {
            const uint32_t kt = counts[t] > 1 ? counts[t] : 1;
            place += (kt > key || (kt == key && t < tid)) ? 1u : 0u;
            k += counts[t] > 1;
        }
        order[place] = (uint16_t)tid;
        if (tid == 0) *ncounted = k;
    }
};
// all 512 SymRankCoders start from that order (src/lz.rs:259-263, src/symrank.rs:22-36): thread = (ctx, rank)
struct CensusFill {
    const uint16_t* order;
    uint16_t* srstate;  // [512][kSrWords]
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t c = (uint32_t)(tid / kSyms), i = (uint32_t)(tid % kSyms);
        if (c >= 512) return;
        uint16_t* t = srstate + (size_t)c * kSrWords;
        const uint16_t v = order[i];
        t[i] = v;
        t[kSyms + v] = (uint16_t)i;
        if (i == 0) {
            const uint32_t cnt = 0, sum = 1000000;  // src/symrank.rs:26-27
            t[2 * kSyms + 0] = (uint16_t)cnt; t[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
            t[2 * kSyms + 2] = (uint16_t)sum; t[2 * kSyms + 3] = (uint16_t)(sum >> 16);
        }
    }
};

// SymRankCoder::encode + update on tables held in fast memory (LDS on the GPU)
// `magic` (optional): magic[d] = floor(2^32 / d) + 1 for d in [2, 391]; then floor(n / d) == mulhi(n, magic[d])
// for every n < 2^32 / d -- here n = sum / 16 < 2^17 and d = cnt <= 390 (checked exhaustively in tests/).
ORZ_HD uint16_t symrank_encode(uint16_t* value, uint16_t* index, uint32_t& cnt, uint32_t& sum, uint16_t v,
                               uint16_t vun, const uint32_t* magic = nullptr) {
    uint16_t i = index[v];
    uint16_t iu = index[vun];
    if (cnt > kSyms) {  // src/symrank.rs:63-66
        cnt = cnt * 9 / 10;
        sum = sum * 9 / 10;
    }
    cnt += 1;
    sum += i;
    const uint32_t n16 = sum / 16;
    const uint32_t q = (magic && cnt > 1) ? (uint32_t)(((uint64_t)n16 * magic[cnt]) >> 32) : n16 / cnt;
    uint16_t dec = (uint16_t)(i / 16 + (uint16_t)q);
    uint16_t next_i = i > dec ? (uint16_t)(i - dec) : 0;
    if (next_i < i / 2) next_i = i / 2;
    uint16_t n = i - next_i;
    if (n == 1) {
        uint16_t nv1 = value[next_i];
        index[v] = next_i;
        value[i] = nv1;
        index[nv1] = i;
        value[next_i] = v;
    } else if (n > 1) {
        uint16_t ni2 = next_i, ni1 = next_i + n / 2;
        uint16_t nv1 = value[ni1], nv2 = value[ni2];
        value[i] = nv1;
        index[nv1] = i;
        value[ni1] = nv2;
        index[nv2] = ni1;
        value[ni2] = v;
        index[v] = ni2;
    }
    if (i == iu) return kSyms - 1;
    return i - (i > iu);
}

// The items of each context side by side: a stable sort of the item indices by context (9 key bits; the backend's
// sort_by_ctx) gives `perm` (sorted slot -> item) and `skey` (the sorted contexts).
struct SymGather {  // contiguous (symbol | unlikely << 16) stream per context
    const uint32_t* perm;
    const uint16_t* isym;
    const uint8_t* iunl;
    uint32_t n;
    uint32_t* gsym;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        const uint32_t i = perm[tid];
        gsym[tid] = isym[i] | ((uint32_t)iunl[i] << 16);
    }
};
struct SymRunStart {  // rstart[c] = first sorted slot with ctx >= c (binary search), c in [0,512]
    const uint16_t* skey;
    uint32_t n;
    uint32_t* rstart;
    ORZ_HD void operator()(size_t tid) const {
        if (tid > 512) return;
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            uint32_t mid = (lo + hi) / 2;
            if (skey[mid] < tid) lo = mid + 1; else hi = mid;
        }
        rstart[tid] = lo;
    }
};
// body of the symrank kernel for one context; `value`/`index` point to fast memory
ORZ_HD void symrank_run(uint16_t* value, uint16_t* index, uint16_t* state, const uint32_t* gsym, uint16_t* grank,
                        uint32_t a, uint32_t e) {
    if (a >= e) return;
    for (uint32_t i = 0; i < kSyms; i++) { value[i] = state[i]; index[i] = state[kSyms + i]; }
    uint32_t cnt = state[2 * kSyms] | ((uint32_t)state[2 * kSyms + 1] << 16);
    uint32_t sum = state[2 * kSyms + 2] | ((uint32_t)state[2 * kSyms + 3] << 16);
    for (uint32_t j = a; j < e; j++) {
        uint32_t g = gsym[j];
        grank[j] = symrank_encode(value, index, cnt, sum, (uint16_t)(g & 0xffff), (uint16_t)(g >> 16));
    }
    for (uint32_t i = 0; i < kSyms; i++) { state[i] = value[i]; state[kSyms + i] = index[i]; }
    state[2 * kSyms] = (uint16_t)cnt; state[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
    state[2 * kSyms + 2] = (uint16_t)sum; state[2 * kSyms + 3] = (uint16_t)(sum >> 16);
}
struct SymScatter {
    const uint32_t* perm;
    const uint16_t* grank;
    uint32_t n;
    uint16_t* irank;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) irank[perm[tid]] = grank[tid];
    }
};

// ---------------------------------------------------------------------------------------------
// per-chunk Huffman weights (src/lz.rs:272-305).  hw layout per chunk: [389 | 389 | 240] u32
constexpr uint32_t kHwStride = kSyms * 2 + kLenSyms;
struct Hist {
    const uint16_t* irank;
    const uint8_t* ial;
    const uint8_t* ienc;
    uint32_t n;
    uint32_t* hw;  // [nchunks][kHwStride]
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        uint32_t ch = (uint32_t)(tid >> 20);
        uint32_t* w = hw + (size_t)ch * kHwStride;
        uint32_t al = ial[tid] & 1;
        ORZ_ATOMIC_ADD(&w[al * kSyms + irank[tid]], 1u);
        if ((ial[tid] & 2) && ienc[tid] >= 5) ORZ_ATOMIC_ADD(&w[2 * kSyms + ienc[tid]], 1u);
    }
};

// The same histogram, one wavefront per 4096 items (a chunk holds 2^20 items, so a wavefront stays inside one chunk):
// bins privatised in LDS, the non-empty ones flushed with one global atomic each.
struct HistWave {
    const uint16_t* irank;
    const uint8_t* ial;
    const uint8_t* ienc;
    uint32_t n;
    uint32_t* hw;  // [nchunks][kHwStride]
    static size_t lds_bytes() { return kHwStride * 4; }
    template <class W>
    ORZ_HD void operator()(W& w) const {
        uint32_t* bins = (uint32_t*)w.lds();
        const uint32_t lane = w.lane();
        for (uint32_t b = lane; b < kHwStride; b += 64) bins[b] = 0;
        w.sync();
        const size_t base = (size_t)w.block() * 4096;
        for (uint32_t k = 0; k < 64; k++) {
            const size_t tid = base + (size_t)k * 64 + lane;
            if (tid >= n) break;
            const uint32_t al = ial[tid] & 1;
            ORZ_ATOMIC_ADD(&bins[al * kSyms + irank[tid]], 1u);
            if ((ial[tid] & 2) && ienc[tid] >= 5) ORZ_ATOMIC_ADD(&bins[2 * kSyms + ienc[tid]], 1u);
        }
        w.sync();
        uint32_t* dst = hw + (base >> 20) * kHwStride;
        for (uint32_t b = lane; b < kHwStride; b += 64)
            if (bins[b]) ORZ_ATOMIC_ADD(&dst[b], bins[b]);
    }
};

// HuffmanTable::new_from_sym_weights (src/huffman.rs:27-111) + canonical codes (:118-141): the argument block and the
// wave kernel that builds the tables, one wavefront per (chunk, table).
struct HuffBuild {
    const uint32_t* hw;  // [nchunks][kHwStride] symbol weights (< 2^23 each: a chunk has at most 2^20 items)
    uint32_t nchunks;
    uint8_t* hl;         // [nchunks][kHwStride] code lengths
    uint16_t* hc;        // [nchunks][kHwStride] codes
    static constexpr uint32_t kMaxWeight = (1u << 23) - 1;
    ORZ_HD static uint32_t table_off(uint32_t t) { return t == 0 ? 0 : (t == 1 ? kSyms : 2 * kSyms); }
    ORZ_HD static uint32_t table_syms(uint32_t t) { return t == 2 ? kLenSyms : kSyms; }
};

// The reference builds the tree with a min-heap on (weight, index).  Its keys are unique, so the pop order -- hence the
// tree -- is fully determined, and the two-queue construction pops in the same order: the used symbols sorted by
// (weight, index), merged with the queue of internal nodes, whose weights never decrease and whose indices (above every
// symbol's) grow with creation; a leaf wins a weight tie against an internal node (smaller index).
//   sort        every lane ranks its <= 7 symbols by counting smaller keys (weight << 9 | symbol) over the table in LDS
//   merge       serial by definition, <= 388 steps on wave-uniform values: both queues are spread over the lanes of a few
//               registers and a queue head is fetched with v_readlane; leaves are named by their place in the sorted order
//   depths      pointer jumping over the parent links: <= 9 rounds of 13 elements a lane instead of a serial walk
//   length cap  src/huffman.rs:99-108: deeper than 15 -> the leaf weights shrink (cumulatively) and the tree is built again
//   codes       src/huffman.rs:118-141: first code of a length + the symbol's place among the symbols of that length,
//               counted with one ballot per (block of 64 symbols, length)
struct HuffWave {
    HuffBuild f;
    static constexpr uint32_t kSlots = (kSyms + 63) / 64;          // symbols per lane
    static constexpr uint32_t kElems = (2 * kSyms - 1 + 63) / 64;  // tree nodes per lane
    struct Lds {
        uint32_t key[kSyms];    // weight << 9 | symbol of the used symbols, ~0 for the others
        uint32_t sw[kSyms];     // leaf weights in sorted order
        uint16_t par[2 * kSyms], dep[2 * kSyms];  // leaves 0..m-1 by sorted place, internal nodes m..2m-2
        uint32_t first[16];
    };
    static size_t lds_bytes() { return sizeof(Lds); }
    ORZ_HD static uint32_t popc64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__popcll(v);
#else
        return (uint32_t)__builtin_popcountll(v);
#endif
    }
    // slot i (wave-uniform) of a register array without indexing it dynamically
    ORZ_HD static uint32_t slot_get(const uint32_t (&a)[kSlots], uint32_t i) {
        uint32_t v = a[0];
#pragma unroll
        for (uint32_t k = 1; k < kSlots; k++) v = i == k ? a[k] : v;
        return v;
    }
    ORZ_HD static void slot_set(uint32_t (&a)[kSlots], uint32_t i, uint32_t v) {
#pragma unroll
        for (uint32_t k = 0; k < kSlots; k++) a[k] = i == k ? v : a[k];
    }
    template <class W>
    ORZ_D void operator()(W& wv) const {
        Lds& L = *reinterpret_cast<Lds*>(wv.lds());
        const uint32_t tid = wv.block(), ch = tid / 3, t = tid % 3, lane = wv.lane();
        const uint32_t n = HuffBuild::table_syms(t), off = HuffBuild::table_off(t);
        const uint32_t* src = f.hw + (size_t)ch * kHwStride + off;
        uint8_t* lens = f.hl + (size_t)ch * kHwStride + off;
        uint16_t* codes = f.hc + (size_t)ch * kHwStride + off;
        uint32_t w0[kSlots], wc[kSlots];  // original / current weight of symbol lane + 64 k
        uint32_t m = 0;
#pragma unroll
        for (uint32_t k = 0; k < kSlots; k++) {
            const uint32_t s = lane + 64 * k;
            w0[k] = s < n ? src[s] : 0;
            wc[k] = w0[k];
            m += popc64(wv.ballot(w0[k] != 0));  // the ORIGINAL weights decide who takes part (src/huffman.rs:58)
        }
        if (m <= 1) {
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++) {
                const uint32_t s = lane + 64 * k;
                if (s < n) { lens[s] = (m == 1 && w0[k]) ? 1 : 0; codes[s] = 0; }
            }
            return;
        }
        const uint32_t nelem = 2 * m - 1, root = nelem - 1;
        uint32_t r[kSlots], len[kSlots];
        for (;;) {
            uint32_t mykey[kSlots];
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++) {
                const uint32_t s = lane + 64 * k;
                mykey[k] = w0[k] ? (wc[k] << 9) | s : 0xffffffffu;
                if (s < n) L.key[s] = mykey[k];
                r[k] = 0;
            }
            wv.sync();
            for (uint32_t u = 0; u < n; u++) {
                const uint32_t ku = L.key[u];
#pragma unroll
                for (uint32_t k = 0; k < kSlots; k++) r[k] += ku < mykey[k];
            }
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++)
                if (w0[k]) L.sw[r[k]] = wc[k];
            wv.sync();
            // The merge.  Both queues live in registers across the wave -- place q of the sorted leaves (and internal node q in
            // creation order) in lane q & 63 of slot q >> 6 -- so a queue head is one v_readlane away instead of an LDS round
            // trip, and the loop runs on uniform values; only the parent links go to LDS.
            const uint32_t kInf = 0xffffffffu;
            uint32_t swr[kSlots], nwr[kSlots];
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++) {
                swr[k] = lane + 64 * k < m ? L.sw[lane + 64 * k] : kInf;
                nwr[k] = kInf;
            }
            {
                uint32_t qa = 0, qb = 0, made = 0;  // leaves taken, internal nodes taken / made
                uint32_t lcur = swr[0], rcur = kInf, wcur = kInf;  // slots under the leaf head / the node head / the newest node
                uint32_t la = wv.bcast(lcur, 0), lb = kInf;
                for (uint32_t step = 0; step + 1 < m; step++) {
                    uint32_t wsum = 0;
                    for (int pick = 0; pick < 2; pick++) {
                        if (la <= lb) {  // equal weights: the leaf's index is the smaller one
                            if (lane == 0) L.par[qa] = (uint16_t)(m + made);
                            wsum += la;
                            qa++;
                            if ((qa & 63) == 0) lcur = slot_get(swr, qa >> 6);
                            la = wv.bcast(lcur, qa & 63);  // (kInf behind the last leaf)
                        } else {
                            if (lane == 0) L.par[m + qb] = (uint16_t)(m + made);
                            wsum += lb;
                            qb++;
                            if ((qb & 63) == 0) rcur = slot_get(nwr, qb >> 6);
                            const uint32_t from = (qb >> 6) == (made >> 6) ? wcur : rcur;
                            const uint32_t v = wv.bcast(from, qb & 63);
                            lb = qb < made ? v : kInf;
                        }
                    }
                    if (lane == (made & 63)) wcur = wsum;
                    if (qb == made) lb = wsum;  // the queue of internal nodes was empty
                    made++;
                    if ((made & 63) == 0) {  // the slot is full: file it (the node head may still be inside it)
                        slot_set(nwr, (made >> 6) - 1, wcur);
                        if ((qb >> 6) == (made >> 6) - 1) rcur = wcur;
                        wcur = kInf;
                    }
                }
            }
            wv.sync();
            // depth of x = hops to the root: d[x] += d[p[x]], p[x] = p[p[x]] until every link points at the root
            uint32_t pp[kElems], dd[kElems];
#pragma unroll
            for (uint32_t j = 0; j < kElems; j++) {
                const uint32_t x = lane + 64 * j;
                pp[j] = x < root ? L.par[x] : root;
                dd[j] = x < root ? 1 : 0;
            }
            wv.sync();
            for (uint32_t round = 0; round < 10; round++) {
#pragma unroll
                for (uint32_t j = 0; j < kElems; j++) {
                    const uint32_t x = lane + 64 * j;
                    if (x < nelem) { L.par[x] = (uint16_t)pp[j]; L.dep[x] = (uint16_t)dd[j]; }
                }
                wv.sync();
                bool moved = false;
#pragma unroll
                for (uint32_t j = 0; j < kElems; j++) {
                    const uint32_t x = lane + 64 * j;
                    if (x < nelem) {
                        const uint32_t q = pp[j];
                        dd[j] += L.dep[q];
                        pp[j] = L.par[q];
                        moved |= pp[j] != q;
                    }
                }
                const bool more = wv.ballot(moved) != 0;
                wv.sync();
                if (!more) break;
            }
#pragma unroll
            for (uint32_t j = 0; j < kElems; j++) {
                const uint32_t x = lane + 64 * j;
                if (x < nelem) L.dep[x] = (uint16_t)dd[j];
            }
            wv.sync();
            uint32_t cur_max = 0;
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++) {
                len[k] = w0[k] ? L.dep[r[k]] : 0;
                cur_max = len[k] > cur_max ? len[k] : cur_max;
            }
            for (uint32_t o = 32; o; o >>= 1) {
                const uint32_t v = wv.shfl(cur_max, lane ^ o);
                cur_max = v > cur_max ? v : cur_max;
            }
            wv.sync();
            if (cur_max <= 15) break;
            const uint32_t sh = cur_max - 15;  // src/huffman.rs:99-108: weights / 2^sh, at least 1, on top of earlier shrinks
#pragma unroll
            for (uint32_t k = 0; k < kSlots; k++)
                if (wc[k]) { const uint32_t v = wc[k] >> sh; wc[k] = v > 1 ? v : 1; }
        }
        // canonical codes in (length, symbol) order
        uint32_t cnt[16], rnk[kSlots];
#pragma unroll
        for (uint32_t l = 0; l < 16; l++) cnt[l] = 0;
        const uint64_t below = (1ull << lane) - 1;
#pragma unroll
        for (uint32_t k = 0; k < kSlots; k++) {
            rnk[k] = 0;
#pragma unroll
            for (uint32_t l = 1; l <= 15; l++) {
                const uint64_t b = wv.ballot(len[k] == l);
                if (len[k] == l) rnk[k] = cnt[l] + popc64(b & below);
                cnt[l] += popc64(b);
            }
        }
        if (lane == 0) {
            uint32_t bits = 0, cur = 1;
#pragma unroll
            for (uint32_t l = 1; l <= 15; l++) {
                L.first[l] = 0;
                if (!cnt[l]) continue;
                if (l > cur) { bits <<= (l - cur); cur = l; }
                L.first[l] = bits;
                bits += cnt[l];
            }
        }
        wv.sync();
#pragma unroll
        for (uint32_t k = 0; k < kSlots; k++) {
            const uint32_t s = lane + 64 * k;
            if (s < n) {
                lens[s] = (uint8_t)len[k];
                codes[s] = len[k] ? (uint16_t)(L.first[len[k]] + rnk[k]) : 0;
            }
        }
    }
};

// bits each item occupies (src/lz.rs:320-342)
struct ItemBits {
    const uint16_t* irank;
    const uint8_t* ial;
    const uint8_t* ienc;
    const uint16_t* irob;
    const uint8_t* hl;
    uint32_t n;
    uint32_t* blen;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        const uint8_t* l = hl + (size_t)(tid >> 20) * kHwStride;
        uint32_t al = ial[tid] & 1;
        uint32_t bits = l[al * kSyms + irank[tid]];
        if (ial[tid] & 2) {
            bits += irob[tid] >> 12;
            if (ienc[tid] >= 5) bits += l[2 * kSyms + ienc[tid]];
        }
        blen[tid] = bits;
    }
};

// MSB-first bit writer into a zeroed buffer of big-endian 32-bit words (src/coder.rs:159-217).
// Concurrent writers use OR, so neighbouring items may share a word.
ORZ_HD void put_bits(uint32_t* out, uint64_t bitpos, uint64_t value, uint32_t nbits) {
    while (nbits > 0) {
        uint64_t w = bitpos >> 5;
        uint32_t used = (uint32_t)(bitpos & 31), room = 32 - used;
        uint32_t take = nbits < room ? nbits : room;
        uint32_t chunk = (uint32_t)((value >> (nbits - take)) & ((take == 32) ? 0xffffffffu : ((1u << take) - 1)));
        uint32_t word = chunk << (room - take);
        uint32_t be = ((word & 0xff) << 24) | ((word & 0xff00) << 8) | ((word >> 8) & 0xff00) | (word >> 24);
        ORZ_ATOMIC_OR(&out[w], be);
        bitpos += take;
        nbits -= take;
    }
}
ORZ_HD uint64_t put_varint(uint32_t* out, uint64_t bitpos, uint32_t v) {  // src/coder.rs:27-38
    for (;;) {
        uint32_t has_next = v > 1;
        put_bits(out, bitpos, (v & 1) | (has_next << 1), 2);
        bitpos += 2;
        v >>= 1;
        if (!has_next) break;
    }
    return bitpos;
}
ORZ_HD uint64_t put_table(uint32_t* out, uint64_t bitpos, const uint8_t* lens, uint32_t n) {  // src/coder.rs:45-67
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < n; i++)
        if (lens[i] > maxlen) maxlen = lens[i];
    bitpos = put_varint(out, bitpos, maxlen);
    uint32_t last = 0xffffffffu;
    for (uint32_t sy = 0; sy < n; sy++)
        if (lens[sy] > 0) {
            bitpos = put_varint(out, bitpos, last == 0xffffffffu ? sy + 1 : sy - last);
            bitpos = put_varint(out, bitpos, maxlen - lens[sy]);
            last = sy;
        }
    return put_varint(out, bitpos, 0);
}

// chunk header: [stream start: varint(k), k x u9] varint(end_spos) varint(n_items) T0 T1 T2
struct ChunkHeader {
    const uint8_t* hl;
    uint32_t nchunks, nitems, len;
    uint32_t pos_base;         // bytes of the decoder's block that precede this encoder window's new region (units of a block)
    const uint32_t* ipos;
    const uint16_t* order;     // census order (first chunk of the stream only)
    const uint32_t* ncounted;
    int stream_start;
    uint32_t* out;             // chunk c writes at word offset outoff[c]
    const uint64_t* outoff;    // [nchunks] word offsets
    uint32_t* hdrbits;         // [nchunks] out
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nchunks) return;
        uint32_t* o = out + outoff[tid];
        uint64_t bp = 0;
        if (tid == 0 && stream_start) {  // src/lz.rs:253-256
            uint32_t k = *ncounted;
            bp = put_varint(o, bp, k);
            for (uint32_t i = 0; i < k; i++) { put_bits(o, bp, order[i], 9); bp += 9; }
        }
        uint32_t i0 = (uint32_t)tid << 20;
        uint32_t i1 = i0 + kChunkItems < nitems ? i0 + kChunkItems : nitems;
        uint32_t end_spos = i1 < nitems ? ipos[i1] : len;  // src/lz.rs:268
        bp = put_varint(o, bp, end_spos + pos_base);  // (the position in the DECODER's window)
        bp = put_varint(o, bp, i1 - i0);
        const uint8_t* l = hl + (size_t)tid * kHwStride;
        bp = put_table(o, bp, l, kSyms);
        bp = put_table(o, bp, l + kSyms, kSyms);
        bp = put_table(o, bp, l + 2 * kSyms, kLenSyms);
        hdrbits[tid] = (uint32_t)bp;
    }
};

// The same header, one wavefront per chunk (src/coder.rs:27-67, src/lz.rs:253-270, 311-318).  A table is written by
// all lanes: lane L owns the symbols [7 L, 7 L + 7) -- it needs the last coded symbol before its range (a running
// maximum over the lanes) and the bits written before its range (a running sum), both scanned through LDS; a varint takes
// two bits per payload bit.
ORZ_HD uint32_t varint_bits(uint32_t v) {
    uint32_t g = 1;
    while (v > 1) { v >>= 1; g++; }
    return 2 * g;
}
struct ChunkHeaderWave {
    const uint8_t* hl;
    uint32_t nchunks, nitems, len;
    uint32_t pos_base;
    const uint32_t* ipos;
    const uint16_t* order;
    const uint32_t* ncounted;
    int stream_start;
    uint32_t* out;
    const uint64_t* outoff;
    uint32_t* hdrbits;
    static constexpr uint32_t kPer = 7;  // symbols per lane: 64 * 7 >= 389
    static size_t lds_bytes() { return 64 * 4 * 2 + 16; }
    template <class W>
    ORZ_HD void operator()(W& w) const {
        uint32_t* sBits = (uint32_t*)w.lds();  // per lane: bits of its symbols -> bits before its range
        uint32_t* sLast = sBits + 64;          // per lane: its last coded symbol + 1 (0 = none) -> the last one before its range
        const uint32_t ch = w.block(), lane = w.lane();
        if (ch >= nchunks) return;
        uint32_t* o = out + outoff[ch];
        uint64_t bp = 0;
        if (ch == 0 && stream_start) {  // src/lz.rs:253-256
            const uint32_t k = *ncounted;
            if (lane == 0) put_varint(o, 0, k);
            bp = varint_bits(k);
            for (uint32_t i = lane; i < k; i += 64) put_bits(o, bp + 9ull * i, order[i], 9);
            bp += 9ull * k;
        }
        const uint32_t i0 = ch << 20;
        const uint32_t i1 = i0 + kChunkItems < nitems ? i0 + kChunkItems : nitems;
        const uint32_t end_spos = (i1 < nitems ? ipos[i1] : len) + pos_base;  // src/lz.rs:268 (the position in the DECODER's window)
        if (lane == 0) { const uint64_t b2 = put_varint(o, bp, end_spos); put_varint(o, b2, i1 - i0); }
        bp += varint_bits(end_spos) + varint_bits(i1 - i0);
        for (uint32_t tb = 0; tb < 3; tb++) {
            const uint32_t n = tb < 2 ? kSyms : kLenSyms;
            const uint8_t* lens = hl + (size_t)ch * kHwStride + (tb == 0 ? 0 : tb == 1 ? kSyms : 2 * kSyms);
            // longest code of the table (every lane reads its seven lengths)
            uint8_t mine[kPer];
            uint32_t mx = 0, lastc = 0;
            for (uint32_t k = 0; k < kPer; k++) {
                const uint32_t sy = lane * kPer + k;
                mine[k] = sy < n ? lens[sy] : 0;
                if (mine[k] > mx) mx = mine[k];
                if (mine[k]) lastc = sy + 1;
            }
            sBits[lane] = mx;
            sLast[lane] = lastc;
            w.sync();
            uint32_t maxlen = 0;
            for (uint32_t k = 0; k < 64; k++) if (sBits[k] > maxlen) maxlen = sBits[k];
            uint32_t prev = 0;  // last coded symbol before this lane's range, + 1
            for (uint32_t k = 0; k < lane; k++) if (sLast[k]) prev = sLast[k];
            w.sync();
            // bits of this lane's symbols
            uint32_t nb = 0, pl = prev;
            for (uint32_t k = 0; k < kPer; k++)
                if (mine[k]) {
                    const uint32_t sy = lane * kPer + k;
                    nb += varint_bits(pl ? sy + 1 - pl : sy + 1) + varint_bits(maxlen - mine[k]);
                    pl = sy + 1;
                }
            sBits[lane] = nb;
            w.sync();
            uint32_t before = 0, total = 0;
            for (uint32_t k = 0; k < 64; k++) { if (k < lane) before += sBits[k]; total += sBits[k]; }
            w.sync();
            if (lane == 0) put_varint(o, bp, maxlen);
            const uint64_t tbase = bp + varint_bits(maxlen);
            uint64_t at = tbase + before;
            pl = prev;
            for (uint32_t k = 0; k < kPer; k++)
                if (mine[k]) {
                    const uint32_t sy = lane * kPer + k;
                    at = put_varint(o, at, pl ? sy + 1 - pl : sy + 1);
                    at = put_varint(o, at, maxlen - mine[k]);
                    pl = sy + 1;
                }
            if (lane == 0) put_varint(o, tbase + total, 0);
            bp = tbase + total + 2;
        }
        if (lane == 0) hdrbits[ch] = (uint32_t)bp;
    }
};

// The census of the stream's first chunk (src/lz.rs:240-246), one wavefront per 4096 items: bins privatised in LDS
struct CensusCountWave {
    const uint16_t* isym;
    uint32_t n;
    uint32_t* counts;
    static size_t lds_bytes() { return (kSyms + 3) * 4; }
    template <class W>
    ORZ_HD void operator()(W& w) const {
        uint32_t* bins = (uint32_t*)w.lds();
        const uint32_t lane = w.lane();
        for (uint32_t b = lane; b < kSyms; b += 64) bins[b] = 0;
        w.sync();
        const size_t base = (size_t)w.block() * 4096;
        for (uint32_t k = 0; k < 64; k++) {
            const size_t tid = base + (size_t)k * 64 + lane;
            if (tid >= n) break;
            ORZ_ATOMIC_ADD(&bins[isym[tid]], 1u);
        }
        w.sync();
        for (uint32_t b = lane; b < kSyms; b += 64)
            if (bins[b]) ORZ_ATOMIC_ADD(&counts[b], bins[b]);
    }
};

struct Pack {  // src/lz.rs:320-342
    const uint16_t* irank;
    const uint8_t* ial;
    const uint8_t* ienc;
    const uint16_t* irob;
    const uint8_t* hl;
    const uint16_t* hc;
    const uint32_t* bscan;    // exclusive scan of blen over all items of the block
    const uint32_t* hdrbits;
    const uint64_t* outoff;
    uint32_t n;
    uint32_t* out;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        uint32_t ch = (uint32_t)(tid >> 20);
        const uint8_t* l = hl + (size_t)ch * kHwStride;
        const uint16_t* c = hc + (size_t)ch * kHwStride;
        uint64_t bp = (uint64_t)hdrbits[ch] + (bscan[tid] - bscan[(size_t)ch << 20]);
        uint32_t* o = out + outoff[ch];
        uint32_t al = ial[tid] & 1, r = irank[tid];
        uint32_t nb = l[al * kSyms + r];
        put_bits(o, bp, c[al * kSyms + r], nb);
        bp += nb;
        if (ial[tid] & 2) {
            uint32_t rl = irob[tid] >> 12;
            put_bits(o, bp, irob[tid] & 0xfff, rl);
            bp += rl;
            uint32_t e = ienc[tid];
            if (e >= 5) put_bits(o, bp, c[2 * kSyms + e], l[2 * kSyms + e]);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// block end: word-predictor snapshot (the table is byte values, it never expires: src/lz.rs:52,203,233)
struct WordsApply {
    const uint8_t* win;
    const uint32_t* wlast;
    uint32_t len;
    uint32_t last_type;  // type of the block's last item
    uint8_t* wsnap;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= 32768) return;
        uint32_t u = wlast[tid];
        if (u) { wsnap[tid * 2] = win[u]; wsnap[tid * 2 + 1] = win[u + 1]; }
        if (last_type != kTyWord && hash2(win, len - 3) == tid) {
            wsnap[tid * 2] = win[len - 2];
            wsnap[tid * 2 + 1] = win[len - 1];
        }
    }
};

// window slide by `sh` positions for per-position arrays, in place: a[x] = a[x + sh] for x in [off, off + sh) -- the
// caller walks off upwards so that a launch never reads what it writes; slot 0 is invalid
template <class T>
struct SlideArray {
    T* a;
    uint32_t off, sh, end;  // end: first position not to write (kPre)
    ORZ_HD void operator()(size_t tid) const {
        const uint32_t x = off + (uint32_t)tid;
        if (tid >= sh || x >= end) return;
        a[x] = x == 0 ? (T)0 : a[x + sh];
    }
};
}  // namespace orz
