// orz_kernels.h -- kernel bodies of the MI355X ROLZ encoder (one functor per kernel).
//
// Each functor's operator()(tid) is the per-thread body; the HIP backend launches it as a
// __global__ grid on gfx950, the emulation backend (tests only) runs it in a host loop.
//
// What replaces what (reference = /root/reference, Rust):
//   ParseSeg      LZEncoder::encode parse loop src/lz.rs:131-235 + BucketMatcher::find_match /
//                 has_lazy_match src/matcher.rs:135-228 + Bucket/BucketMatcher::update :62-80,115-121,
//                 re-stated as a speculative segment-parallel fixed-point iteration
//   Rank*         ring ordinals = what Bucket.head / node_size_bounded_sub encode, src/matcher.rs:62-91
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
#else  // host emulation runs the threads of a kernel one after another
template <class T, class U> inline T orz_fetch_add(T* p, U v) { T o = *p; *p = (T)(o + v); return o; }
template <class T, class U> inline T orz_fetch_min(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T orz_fetch_max(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T orz_fetch_or(T* p, U v) { T o = *p; *p = (T)(o | v); return o; }
#define ORZ_ATOMIC_ADD(p, v) orz_fetch_add((p), (v))
#define ORZ_ATOMIC_MIN(p, v) orz_fetch_min((p), (v))
#define ORZ_ATOMIC_MAX(p, v) orz_fetch_max((p), (v))
#define ORZ_ATOMIC_OR(p, v) orz_fetch_or((p), (v))
#endif

namespace orz {

// ---------------------------------------------------------------------------------------------
// K0: per-position keys.  Builds the two sort inputs of a block:
//   ent[j]  = bucket_key(x) << 25 | x   for history item starts x in [1,P) and all x in [P,len)
//   kent[j] = hash2(u-1)    << 25 | u   for u in [P-1,len)       (word-predictor chain)
struct BuildEntries {
    const uint8_t* win;     // window base, win[-480 .. kBlock+480) readable
    const uint32_t* hpos;   // compacted history item starts (ascending), nhist entries
    uint32_t nhist, n;      // n new bytes
    uint64_t* ent;          // nhist + n
    uint64_t* kent;         // n + 1
    ORZ_HD void operator()(size_t tid) const {
        if (tid < nhist) {
            uint32_t x = hpos[tid];
            ent[tid] = ((uint64_t)bucket_key(win, x) << kPosBits) | x;
        } else if (tid < (size_t)nhist + n) {
            uint32_t x = kPre + (uint32_t)(tid - nhist);
            ent[tid] = ((uint64_t)bucket_key(win, x) << kPosBits) | x;
        }
        if (tid < (size_t)n + 1) {
            uint32_t u = kPre - 1 + (uint32_t)tid;
            kent[tid] = ((uint64_t)hash2(win, u - 1) << kPosBits) | u;
        }
    }
};

// after sorting: idx[x] = slot of in-block position x in ent; kidx[u] likewise for kent
struct ScatterIndex {
    const uint64_t* ent;
    uint32_t nent;
    uint32_t* idx;  // indexed by window offset
    uint32_t lo;    // only positions >= lo are scattered
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nent) return;
        uint32_t x = (uint32_t)(ent[tid] & kPosMask);
        if (x >= lo) idx[x] = (uint32_t)tid;
    }
};

// ---------------------------------------------------------------------------------------------
// Parse state of one stream on the device (window offsets index every per-position array).
struct ParseState {
    const uint8_t* win;
    uint32_t len;        // kPre + n
    uint32_t seg_size;   // bytes per segment (>= 256, multiple of 8)
    uint32_t nseg;
    const uint64_t* ent;
    const uint32_t* idx;
    const uint64_t* kent;
    const uint32_t* kidx;
    const uint8_t* wsnap;  // words table at block start, [32768][2]
    // double-buffered speculative state (old = previous sweep, new = this sweep)
    const uint8_t* S_old;
    const uint8_t* E_old;
    const uint8_t* ML_old;
    const uint32_t* ORD_old;
    uint8_t* S_new;
    uint8_t* E_new;
    uint8_t* ML_new;
    uint32_t* ORD_new;
    const uint32_t* first_old;
    const uint8_t* lt_old;
    uint32_t* first_new;
    uint8_t* lt_new;
    // single-buffered per-item outputs
    uint16_t* LR;    // rank of the item among same-ctx items of its segment
    uint32_t* SRC;   // match source position
    uint8_t* W0;     // words[hash2(p-1)][0] at the item start (symrank_unlikely)
    uint8_t* TY;     // type | after_literal << 2
    const uint32_t* base;  // [nseg+1][256] items of ctx c before segment sg (stream ordinals)
    uint32_t* first_changed;
    uint8_t lt0;     // last item type before the block (after_literal carry)
    Cfg cfg;
};

// One lane re-parses one segment with the reference's exact decision rules, reading the previous
// sweep's state for everything outside its own segment.  See DESIGN.md section 3 for why the
// fixed point of this iteration is the reference's serial parse.
struct ParseSeg {
    ParseState s;
    uint32_t seg0, seg1;  // evaluate segments [seg0, seg1)
    uint16_t* lcnt;       // [256][seg1-seg0] lane-private per-ctx counters

    struct Cand {
        uint32_t q, ord, valid;
    };

    ORZ_HD void operator()(size_t tid) const {
        uint32_t sg = seg0 + (uint32_t)tid;
        if (sg >= seg1) return;
        const uint8_t* b = s.win;
        const uint32_t seg_start = kPre + sg * s.seg_size;
        const uint32_t seg_end = (seg_start + s.seg_size < s.len) ? seg_start + s.seg_size : s.len;
        // per-ctx count of this lane's own items so far (lane-private column of a global scratch)
        const size_t nl = seg1 - seg0;
        uint16_t* lc = lcnt + tid;
        for (uint32_t c = 0; c < 256; c++) lc[c * nl] = 0;
        for (uint32_t x = seg_start; x < seg_end; x++) { s.S_new[x] = 0; s.E_new[x] = 0; s.ML_new[x] = 0; }
        uint32_t p = sg == 0 ? kPre : s.first_old[sg];
        uint32_t lt = sg == 0 ? s.lt0 : s.lt_old[sg];
        if (sg > 0 && p < seg_end) s.E_new[p] = (lt != kTyWord);

        while (p < seg_end) {
            const uint32_t c = hash1(b, p - 1);
            // ---- word predictor lookup: latest in-block update of key hash2(p-1), else snapshot
            uint32_t kk = hash2(b, p - 1);
            uint8_t w0 = s.wsnap[kk * 2], w1 = s.wsnap[kk * 2 + 1];
            {
                uint32_t j = s.kidx[p];
                while (j > 0) {
                    j--;
                    uint64_t e = s.kent[j];
                    if ((uint32_t)(e >> kPosBits) != kk) break;
                    uint32_t u = (uint32_t)(e & kPosMask);
                    if (u + 2 > p) continue;
                    uint32_t en = u + 2;
                    uint8_t eb = en >= seg_start ? s.E_new[en] : s.E_old[en];
                    if (eb) { w0 = b[u]; w1 = b[u + 1]; break; }
                }
            }
            const uint32_t lwm = (b[p] == w0 && b[p + 1] == w1);
            // ---- find_match (src/matcher.rs:135-192)
            const uint32_t hcnt = s.base[(size_t)sg * 256 + c] + lc[c * nl];
            uint32_t max_len = kMinLen - 1, mlexp = kMinLen, bestq = 0, besto = 0;
            uint32_t mld = ld32(b + p + max_len - 3);
            {
                const uint32_t key = bucket_key(b, p);
                uint32_t j = s.idx[p];
                int cnt = 0;
                while (j > 0 && cnt < s.cfg.depth) {
                    j--;
                    uint64_t e = s.ent[j];
                    if ((uint32_t)(e >> kPosBits) != key) break;
                    uint32_t q = (uint32_t)(e & kPosMask);
                    uint8_t sv = q >= seg_start ? s.S_new[q] : s.S_old[q];
                    if (!sv) continue;
                    uint32_t oq = q >= seg_start ? s.ORD_new[q] : s.ORD_old[q];
                    if (hcnt - 1 - oq > kRing - 1) break;  // fell out of the 4094-entry ring
                    cnt++;
                    if (ld32(b + q + max_len - 3) == mld) {
                        uint32_t l = lcp240(b, q, p);
                        if (l > max_len) {
                            mlexp = q >= seg_start ? s.ML_new[q] : s.ML_old[q];
                            max_len = l;
                            bestq = q;
                            besto = oq;
                            mld = ld32(b + p + max_len - 3);
                        }
                        if (l == kMaxLen) break;
                        if (mlexp > 0 && l > mlexp) break;
                    }
                }
            }
            const bool is_match = max_len >= kMinLen && p + max_len < s.len;
            uint32_t lazy = 0;
            if (is_match && max_len < kMaxLen / 2) {  // src/lz.rs:151-170
                const uint32_t ro = hcnt - 1 - besto;
                const uint32_t l1 = max_len + 1 + (roid_bitlen(ro) < 8), l2 = l1 - lwm;
                if (has_lazy(seg_start, sg, p, p + 1, l1, s.cfg.lazy1, lc, nl)) lazy = 1;
                else if (has_lazy(seg_start, sg, p, p + 2, l2, s.cfg.lazy2, lc, nl)) lazy = 2;
            }
            // ---- commit the item (src/lz.rs:172-234)
            s.S_new[p] = 1;
            s.ORD_new[p] = hcnt;
            s.LR[p] = lc[c * nl];
            lc[c * nl]++;
            s.W0[p] = w0;
            const uint8_t al = (lt == kTyLit) ? 4 : 0;
            if (is_match && !lazy) {
                s.ML_new[p] = (uint8_t)max_len;
                s.SRC[p] = bestq;
                s.TY[p] = kTyMatch | al;
                p += max_len;
                lt = kTyMatch;
            } else if (p + 1 < s.len && lazy != 1 && lwm) {
                s.TY[p] = kTyWord | al;
                p += 2;
                lt = kTyWord;
            } else {
                s.TY[p] = kTyLit | al;
                p += 1;
                lt = kTyLit;
            }
            if (p < seg_end) s.E_new[p] = (lt != kTyWord);
        }
        s.first_new[sg + 1] = p;
        s.lt_new[sg + 1] = (uint8_t)lt;
        // ---- change detection against the previous sweep
        bool ch = s.first_new[sg + 1] != s.first_old[sg + 1] || s.lt_new[sg + 1] != s.lt_old[sg + 1];
        for (uint32_t x = seg_start; x < seg_end && !ch; x++)
            ch = s.S_new[x] != s.S_old[x] || s.ML_new[x] != s.ML_old[x] || s.E_new[x] != s.E_old[x];
        if (ch) ORZ_ATOMIC_MIN(s.first_changed, sg);
    }

    // has_lazy_match (src/matcher.rs:194-228) for the probe position x in {p+1, p+2}: candidates
    // are the items inserted before p, i.e. positions < p.
    ORZ_HD bool has_lazy(uint32_t seg_start, uint32_t sg, uint32_t p, uint32_t x, uint32_t min_len,
                         int depth, const uint16_t* lc, size_t nl) const {
        const uint8_t* b = s.win;
        const uint32_t cx = hash1(b, x - 1);
        const uint32_t hx = s.base[(size_t)sg * 256 + cx] + lc[cx * nl];
        const uint32_t key = bucket_key(b, x);
        uint32_t j = s.idx[x];
        int cnt = 0;
        while (j > 0 && cnt < depth) {
            j--;
            uint64_t e = s.ent[j];
            if ((uint32_t)(e >> kPosBits) != key) break;
            uint32_t q = (uint32_t)(e & kPosMask);
            if (q >= p) continue;
            uint8_t sv = q >= seg_start ? s.S_new[q] : s.S_old[q];
            if (!sv) continue;
            uint32_t oq = q >= seg_start ? s.ORD_new[q] : s.ORD_old[q];
            if (hx - 1 - oq > kRing - 1) break;
            cnt++;
            if (lcp240(b, q, x) >= min_len) return true;  // == mem_fast_equal over min_len bytes
        }
        return false;
    }
};

// ---------------------------------------------------------------------------------------------
// Rank kernels: per-segment per-ctx item histograms -> base table -> stream ordinals.
struct RankHist {
    const uint8_t* win;
    const uint8_t* S;
    uint32_t x0, x1, seg_size;
    uint32_t* hist;  // [nseg][256]
    ORZ_HD void operator()(size_t tid) const {
        uint32_t x = x0 + (uint32_t)tid;
        if (x >= x1 || !S[x]) return;
        uint32_t sg = (x - kPre) / seg_size;
        ORZ_ATOMIC_ADD(&hist[(size_t)sg * 256 + hash1(win, x - 1)], 1u);
    }
};
// two-level scan over segments for each of the 256 contexts
struct RankChunkSum {  // thread = (chunk, c)
    const uint32_t* hist;
    uint32_t seg0, seg1, chunk;  // chunk = segments per chunk
    uint32_t* csum;              // [nchunks][256]
    ORZ_HD void operator()(size_t tid) const {
        uint32_t c = (uint32_t)(tid & 255), ch = (uint32_t)(tid >> 8);
        uint32_t a = seg0 + ch * chunk;
        if (a >= seg1) return;
        uint32_t e = a + chunk < seg1 ? a + chunk : seg1;
        uint32_t sum = 0;
        for (uint32_t sg = a; sg < e; sg++) sum += hist[(size_t)sg * 256 + c];
        csum[(size_t)ch * 256 + c] = sum;
    }
};
struct RankChunkScan {  // thread = c ; exclusive scan of chunk sums, seeded by base[seg0][c]
    uint32_t* csum;
    const uint32_t* base;
    uint32_t seg0, nchunks;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= 256) return;
        uint32_t run = base[(size_t)seg0 * 256 + tid];
        for (uint32_t ch = 0; ch < nchunks; ch++) {
            uint32_t v = csum[(size_t)ch * 256 + tid];
            csum[(size_t)ch * 256 + tid] = run;
            run += v;
        }
    }
};
struct RankApply {  // thread = (chunk, c): base[sg+1] = base[sg] + hist[sg]
    const uint32_t* hist;
    const uint32_t* csum;
    uint32_t seg0, seg1, chunk;
    uint32_t* base;
    ORZ_HD void operator()(size_t tid) const {
        uint32_t c = (uint32_t)(tid & 255), ch = (uint32_t)(tid >> 8);
        uint32_t a = seg0 + ch * chunk;
        if (a >= seg1) return;
        uint32_t e = a + chunk < seg1 ? a + chunk : seg1;
        uint32_t run = csum[(size_t)ch * 256 + c];
        for (uint32_t sg = a; sg < e; sg++) {
            run += hist[(size_t)sg * 256 + c];
            base[(size_t)(sg + 1) * 256 + c] = run;
        }
    }
};
struct RankOrd {  // ORD[x] = base[seg(x)][ctx(x)] + LR[x]
    const uint8_t* win;
    const uint8_t* S;
    const uint16_t* LR;
    const uint32_t* base;
    uint32_t x0, x1, seg_size;
    uint32_t* ORD;
    ORZ_HD void operator()(size_t tid) const {
        uint32_t x = x0 + (uint32_t)tid;
        if (x >= x1 || !S[x]) return;
        uint32_t sg = (x - kPre) / seg_size;
        ORD[x] = base[(size_t)sg * 256 + hash1(win, x - 1)] + LR[x];
    }
};

// ---------------------------------------------------------------------------------------------
// Post-parse: items
struct ItemPos {  // scatter item start positions by their exclusive-scan index
    const uint8_t* S;
    const uint32_t* scan;  // exclusive scan of S over [kPre, len), indexed from 0
    uint32_t n;
    uint32_t* ipos;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        if (S[kPre + tid]) ipos[scan[tid]] = kPre + (uint32_t)tid;
    }
};

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
struct CensusInit {  // one thread: stable order by count (src/lz.rs:247-263); fills all 512 tables
    const uint32_t* counts;
    uint16_t* order;     // [389] out: initial rank order
    uint32_t* ncounted;  // out
    uint16_t* srstate;   // [512][kSrWords]
    ORZ_HD void operator()(size_t tid) const {
        if (tid != 0) return;
        uint16_t vs[kSyms];
        uint32_t k = 0;
        for (uint32_t sy = 0; sy < kSyms; sy++) {
            uint32_t key = counts[sy] > 1 ? counts[sy] : 1;
            k += counts[sy] > 1;
            uint32_t j = sy;
            while (j > 0) {
                uint32_t kp = counts[vs[j - 1]] > 1 ? counts[vs[j - 1]] : 1;
                if (kp >= key) break;
                vs[j] = vs[j - 1];
                j--;
            }
            vs[j] = (uint16_t)sy;
        }
        *ncounted = k;
        for (uint32_t i = 0; i < kSyms; i++) order[i] = vs[i];
        for (uint32_t c = 0; c < 512; c++) {
            uint16_t* t = srstate + (size_t)c * kSrWords;
            for (uint32_t i = 0; i < kSyms; i++) {
                t[i] = vs[i];
                t[kSyms + vs[i]] = (uint16_t)i;
            }
            uint32_t cnt = 0, sum = 1000000;  // src/symrank.rs:26-27
            t[2 * kSyms + 0] = (uint16_t)cnt; t[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
            t[2 * kSyms + 2] = (uint16_t)sum; t[2 * kSyms + 3] = (uint16_t)(sum >> 16);
        }
    }
};

// SymRankCoder::encode + update on tables held in fast memory (LDS on the GPU)
ORZ_HD uint16_t symrank_encode(uint16_t* value, uint16_t* index, uint32_t& cnt, uint32_t& sum, uint16_t v,
                               uint16_t vun) {
    uint16_t i = index[v];
    uint16_t iu = index[vun];
    if (cnt > kSyms) {  // src/symrank.rs:63-66
        cnt = cnt * 9 / 10;
        sum = sum * 9 / 10;
    }
    cnt += 1;
    sum += i;
    uint16_t dec = (uint16_t)(i / 16 + (uint16_t)(sum / 16 / cnt));
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

struct SymKeys {  // key = ctx << 24 | item index
    const uint16_t* ictx;
    uint32_t n;
    uint64_t* keys;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) keys[tid] = ((uint64_t)ictx[tid] << 24) | tid;
    }
};
struct SymGather {  // contiguous (symbol | unlikely << 16) stream per context
    const uint64_t* keys;
    const uint16_t* isym;
    const uint8_t* iunl;
    uint32_t n;
    uint32_t* gsym;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        uint32_t i = (uint32_t)(keys[tid] & 0xffffff);
        gsym[tid] = isym[i] | ((uint32_t)iunl[i] << 16);
    }
};
struct SymRunStart {  // rstart[c] = first sorted slot with ctx >= c (binary search), c in [0,512]
    const uint64_t* keys;
    uint32_t n;
    uint32_t* rstart;
    ORZ_HD void operator()(size_t tid) const {
        if (tid > 512) return;
        uint64_t want = (uint64_t)tid << 24;
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            uint32_t mid = (lo + hi) / 2;
            if (keys[mid] < want) lo = mid + 1; else hi = mid;
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
    const uint64_t* keys;
    const uint16_t* grank;
    uint32_t n;
    uint16_t* irank;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) irank[(uint32_t)(keys[tid] & 0xffffff)] = grank[tid];
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

// HuffmanTable::new_from_sym_weights (src/huffman.rs:27-111) + canonical codes (:118-141).
// One thread per (chunk, table).  The heap is a binary min-heap on (weight, index): keys are
// unique, so the pop order -- hence the tree -- is fully determined.
struct HuffBuild {
    const uint32_t* hw;  // [nchunks][kHwStride]
    uint32_t nchunks;
    uint8_t* hl;         // [nchunks][kHwStride] code lengths
    uint16_t* hc;        // [nchunks][kHwStride] codes
    uint32_t* scratch;   // [nchunks*3][kHuffScratch] u32
    static constexpr uint32_t kHuffScratch = 4 * 2 * kSyms + kSyms;

    ORZ_HD static bool less(uint32_t wa, uint32_t ia, uint32_t wb, uint32_t ib) {
        return wa < wb || (wa == wb && ia < ib);
    }
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= (size_t)nchunks * 3) return;
        uint32_t ch = (uint32_t)(tid / 3), t = (uint32_t)(tid % 3);
        uint32_t off = t == 0 ? 0 : (t == 1 ? kSyms : 2 * kSyms);
        uint32_t n = t == 2 ? kLenSyms : kSyms;
        const uint32_t* w0 = hw + (size_t)ch * kHwStride + off;
        uint8_t* lens = hl + (size_t)ch * kHwStride + off;
        uint16_t* codes = hc + (size_t)ch * kHwStride + off;
        uint32_t* sc = scratch + tid * kHuffScratch;
        uint32_t* w = sc;                  // [2n] node weights
        uint32_t* c1 = sc + 2 * kSyms;     // [2n]
        uint32_t* c2 = sc + 4 * kSyms;     // [2n]
        uint32_t* cl = sc + 6 * kSyms;     // [2n] depths
        uint32_t* heap = sc + 8 * kSyms;   // [n] node ids
        for (uint32_t i = 0; i < n; i++) w[i] = w0[i];
        for (;;) {
            uint32_t hn = 0, nodes = n;
            for (uint32_t i = 0; i < n; i++)
                if (w0[i] > 0) {  // filter on the ORIGINAL weights (src/huffman.rs:58)
                    uint32_t k = hn++;
                    heap[k] = i;
                    while (k > 0) {
                        uint32_t pa = (k - 1) / 2;
                        if (!less(w[heap[k]], heap[k], w[heap[pa]], heap[pa])) break;
                        uint32_t tmp = heap[k]; heap[k] = heap[pa]; heap[pa] = tmp;
                        k = pa;
                    }
                }
            if (hn <= 1) {
                for (uint32_t i = 0; i < n; i++) { lens[i] = 0; codes[i] = 0; }
                if (hn == 1) lens[heap[0]] = 1;
                return;
            }
            while (hn > 1) {
                uint32_t pick[2];
                for (int r = 0; r < 2; r++) {
                    pick[r] = heap[0];
                    heap[0] = heap[--hn];
                    uint32_t k = 0;
                    for (;;) {
                        uint32_t l = 2 * k + 1, rr = l + 1, m = k;
                        if (l < hn && less(w[heap[l]], heap[l], w[heap[m]], heap[m])) m = l;
                        if (rr < hn && less(w[heap[rr]], heap[rr], w[heap[m]], heap[m])) m = rr;
                        if (m == k) break;
                        uint32_t tmp = heap[k]; heap[k] = heap[m]; heap[m] = tmp;
                        k = m;
                    }
                }
                w[nodes] = w[pick[0]] + w[pick[1]];
                c1[nodes] = pick[0];
                c2[nodes] = pick[1];
                uint32_t k = hn++;
                heap[k] = nodes;
                nodes++;
                while (k > 0) {
                    uint32_t pa = (k - 1) / 2;
                    if (!less(w[heap[k]], heap[k], w[heap[pa]], heap[pa])) break;
                    uint32_t tmp = heap[k]; heap[k] = heap[pa]; heap[pa] = tmp;
                    k = pa;
                }
            }
            for (uint32_t i = 0; i < nodes; i++) cl[i] = 0;
            for (uint32_t i = nodes; i-- > n;) {
                cl[c1[i]] = cl[i] + 1;
                cl[c2[i]] = cl[i] + 1;
            }
            uint32_t cur_max = 0;
            for (uint32_t i = 0; i < n; i++)
                if (cl[i] > cur_max) cur_max = cl[i];
            if (cur_max > 15) {  // src/huffman.rs:99-108: cumulative shrink of leaf weights
                uint32_t shrink = 1u << (cur_max - 15);
                for (uint32_t i = 0; i < n; i++)
                    if (w[i] > 0) {
                        uint32_t v = w[i] / shrink;
                        w[i] = v > 1 ? v : 1;
                    }
                continue;
            }
            for (uint32_t i = 0; i < n; i++) lens[i] = (uint8_t)cl[i];
            break;
        }
        // canonical codes in (len, sym) order (src/huffman.rs:118-141)
        uint32_t bits = 0, cur = 1;
        for (uint32_t i = 0; i < n; i++) codes[i] = 0;
        for (uint32_t L = 1; L <= 15; L++)
            for (uint32_t sy = 0; sy < n; sy++) {
                if (lens[sy] != L) continue;
                if (L > cur) { bits <<= (L - cur); cur = L; }
                codes[sy] = (uint16_t)bits;
                bits++;
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
        bp = put_varint(o, bp, end_spos);
        bp = put_varint(o, bp, i1 - i0);
        const uint8_t* l = hl + (size_t)tid * kHwStride;
        bp = put_table(o, bp, l, kSyms);
        bp = put_table(o, bp, l + kSyms, kSyms);
        bp = put_table(o, bp, l + 2 * kSyms, kLenSyms);
        hdrbits[tid] = (uint32_t)bp;
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
struct WordsLast {  // wlast[key] = max u with an update (E bit at u+2)
    const uint8_t* win;
    const uint8_t* E;
    uint32_t len;
    uint32_t* wlast;  // [32768], zeroed
    ORZ_HD void operator()(size_t tid) const {
        uint32_t u = kPre - 1 + (uint32_t)tid;
        if (u + 2 >= len) return;  // the update at e == len is applied by WordsApply (it is the newest)
        if (E[u + 2]) ORZ_ATOMIC_MAX(&wlast[hash2(win, u - 1)], u);
    }
};
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

// window slide for per-position arrays: dst[x] = src[x + 2^24] for x in [0,P), slot 0 invalid
template <class T>
struct SlideArray {
    const T* src;
    T* dst;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= kPre) return;
        dst[tid] = tid == 0 ? (T)0 : src[tid + kNewMax];
    }
};
struct FillFirst {  // segment-start guesses for a fresh block
    uint32_t* a;
    uint32_t* b2;
    uint8_t* la;
    uint8_t* lb;
    uint32_t nseg, seg_size;
    ORZ_HD void operator()(size_t tid) const {
        if (tid > nseg) return;
        uint32_t v = kPre + (uint32_t)tid * seg_size;
        a[tid] = v; b2[tid] = v; la[tid] = kTyLit; lb[tid] = kTyLit;
    }
};
struct IotaFlags {  // history compaction input: flag[x] = S[x] for x in [1,P)
    const uint8_t* S;
    uint8_t* flag;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < kPre) flag[tid] = tid >= 1 && S[tid];
    }
};
struct CompactPos {
    const uint8_t* flag;
    const uint32_t* scan;
    uint32_t n, off;
    uint32_t* out;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n && flag[tid]) out[scan[tid]] = off + (uint32_t)tid;
    }
};

}  // namespace orz
