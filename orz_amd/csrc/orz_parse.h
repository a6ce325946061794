// orz_parse.h -- the ROLZ parse of one stream block as speculative, wave-cooperative sweeps.
//
// What this replaces in the reference (/root/reference, Rust):
//   LZEncoder::encode parse loop            src/lz.rs:131-235
//   BucketMatcher::find_match / has_lazy    src/matcher.rs:135-228   (+ LCP, src/mem.rs:41-70)
//   Bucket::update / BucketMatcher::update  src/matcher.rs:62-80,115-121
//   words[] predictor                       src/lz.rs:132-136,203,233
// The reference runs them as one serial chain over the stream.  Here the block is cut into
// segments of <= 64 positions; one wavefront owns one segment and re-derives its items with the
// reference's exact decision rules, reading every other segment's items as they currently stand
// (speculation).  A sweep evaluates a window of segments beyond the `front` (first non-final
// segment); segments that sit before the first segment whose result changed are final, because
// everything they read was final.  The fixed point is the serial parse.
//
// Data layout (one block, HBM):
//   static, built by the prep kernels once per block
//     epos[slot]      positions sorted by (ctx8, hash_dword % 4627, position)  ("candidate lists")
//     idx[x]          slot of window offset x
//     runstart[key]   first slot of a (ctx8, hash) run
//     kpos/kidx/krun  the same for the word predictor, keyed by hash2 (15 bit)
//   speculative, updated in place by the owning segment's wavefront (only when a value changes)
//     vbits           1 bit per slot: position is an item start (ring member)
//     sml[slot]       the item's match_len_expected (0 literal/word, 255 = not an item)
//     sord[slot]      the item's ordinal in its ctx ring (recomputed per sweep by RankApply)
//     kbits           1 bit per word-predictor slot: words[] was updated at that position + 2
//     exitst[s]       where segment s-1 left the stream: (next item position << 2) | last type
//     hist[s][ctx]    items per context of segment s (ring of R segments) -> base[s][ctx] prefix
#pragma once
#include "orz_common.h"

namespace orz {

constexpr uint32_t kSegMax = 64;            // positions per segment == lanes per wavefront
constexpr uint32_t kNPMax = kSegMax + 2;    // + the two lazy probe positions past the segment
constexpr uint32_t kNoChange = 0xffffffffu;
constexpr uint32_t kRankChunk = 32;         // segments per RankKernel block
constexpr uint32_t kLbPre = 8;              // bytes staged in LDS before the segment start
constexpr uint32_t kLbLen = kLbPre + kNPMax + kMaxLen + 14;  // 330 -> bytes up to x+240+8 readable

struct ParseCtl {           // device-resident sweep control of one stream
    uint32_t front[2];      // first non-final segment, by sweep parity
    uint32_t fchg[2];       // lowest segment whose result changed in the sweep, by parity
    uint32_t evals;         // segments evaluated so far (statistics)
    uint32_t pad[3];
};

struct ParseArgs {
    const uint8_t* win;       // window offset 0 (480 readable bytes before it)
    uint32_t len;             // kPre + n
    uint32_t nseg, seg;       // segments in the block, positions per segment (<= 64)
    uint32_t wsegs;           // segments per sweep window
    uint32_t ring;            // R: rows of the hist/base rings (>= wsegs + 1)
    uint32_t depth, lazy1, lazy2, dmax;  // LZCfg, dmax = max of the three = candidates kept per position
    uint32_t lt0;             // type of the last item of the previous block (after_literal carry)
    uint32_t par;             // sweep parity
    const uint32_t* epos;
    const uint32_t* idx;
    const uint32_t* runstart;
    const uint32_t* kpos;
    const uint32_t* kidx;
    const uint32_t* krun;
    const uint8_t* wsnap;     // words[] as of the block start, [32768][2]
    uint64_t* vbits;
    uint8_t* sml;
    uint32_t* sord;
    uint64_t* kbits;
    uint32_t* exitst;         // [nseg + 1]
    uint8_t* hist;            // [ring][256]
    uint32_t* base;           // [ring][256]
    uint8_t* TY;              // per position outputs of the owning segment
    uint32_t* SRC;
    uint8_t* W0;
    uint8_t* LR;
    uint32_t* partial;        // [2][kRankChunks max][256] per-chunk ctx item counts of the sweep, by parity
    ParseCtl* ctl;
};

#if defined(__HIP_DEVICE_COMPILE__)
ORZ_D uint32_t ldu32(const uint8_t* p) { return *reinterpret_cast<const __attribute__((aligned(1))) uint32_t*>(p); }
ORZ_D uint64_t ldu64(const uint8_t* p) { return *reinterpret_cast<const __attribute__((aligned(1))) uint64_t*>(p); }
ORZ_D void atom_or64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
ORZ_D void atom_and64(uint64_t* p, uint64_t v) { atomicAnd((unsigned long long*)p, (unsigned long long)v); }
ORZ_D void atom_min32(uint32_t* p, uint32_t v) { atomicMin(p, v); }
ORZ_D void atom_add32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
ORZ_D int clz64(uint64_t v) { return __clzll((long long)v); }
ORZ_D int ctz64(uint64_t v) { return __ffsll((long long)v) - 1; }
#else
ORZ_D uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
ORZ_D uint64_t ldu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
ORZ_D void atom_or64(uint64_t* p, uint64_t v) { *p |= v; }
ORZ_D void atom_and64(uint64_t* p, uint64_t v) { *p &= v; }
ORZ_D void atom_min32(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
ORZ_D void atom_add32(uint32_t* p, uint32_t v) { *p += v; }
ORZ_D int clz64(uint64_t v) { return __builtin_clzll(v); }
ORZ_D int ctz64(uint64_t v) { return __builtin_ctzll(v); }
#endif

// common prefix of a[0..) and b[0..), capped at 240, eight bytes a step (== src/mem.rs:41-51)
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

// byte offsets of the per-wave LDS arrays (all sizes for the kNPMax = 66 position case)
struct ParseLds {
    uint32_t lb, idxL, keyL, kkL, ownord, srcL, basec, cq, co;  // u8 / u32 arrays ...
    uint32_t ctxL, ncand, wg, ownv, ownml, ownE, oldml, oldE, tyL, w0L, lrL, cnt, cml, cl, scal, total;
    ORZ_HD static ParseLds make(uint32_t dmax) {
        ParseLds o;
        uint32_t at = 0;
        auto take = [&](uint32_t bytes) { uint32_t r = at; at += (bytes + 7) & ~7u; return r; };
        o.lb = take(kLbLen + 8);
        o.idxL = take(kNPMax * 4);
        o.keyL = take(kNPMax * 4);
        o.kkL = take((kNPMax + 2) * 4);
        o.ownord = take(kNPMax * 4);
        o.srcL = take(kNPMax * 4);
        o.basec = take(256 * 4);
        o.cq = take(kNPMax * dmax * 4);
        o.co = take(kNPMax * dmax * 4);
        o.ctxL = take(kNPMax);
        o.ncand = take(kNPMax);
        o.wg = take(kNPMax * 2);
        o.ownv = take(kNPMax);
        o.ownml = take(kNPMax);
        o.ownE = take(kNPMax);
        o.oldml = take(kNPMax);
        o.oldE = take(kNPMax);
        o.tyL = take(kNPMax);
        o.w0L = take(kNPMax);
        o.lrL = take(kNPMax);
        o.cnt = take(256);
        o.cml = take(kNPMax * dmax);
        o.cl = take(kNPMax * dmax);
        o.scal = take(64);
        o.total = at;
        return o;
    }
};

// ---------------------------------------------------------------------------------------------
// One wavefront = one segment.  W is the wave context: lane(), block(), lds(), ballot(), sync().
struct ParseWave {
    ParseArgs a;

    template <class W>
    ORZ_D void operator()(W& w) const {
        const uint32_t lane = w.lane();
        const uint32_t front = a.ctl->front[a.par];
        const uint32_t sg = front + w.block();
        if (sg >= a.nseg) return;
        const ParseLds L = ParseLds::make(a.dmax);
        uint8_t* lds = w.lds();
        uint8_t* lb = lds + L.lb;  // lb[kLbPre + i] = win[seg_start + i]
        uint32_t* idxL = (uint32_t*)(lds + L.idxL);
        uint32_t* keyL = (uint32_t*)(lds + L.keyL);
        uint32_t* kkL = (uint32_t*)(lds + L.kkL);  // kkL[i] = hash2(seg_start - 2 + i - 1)
        uint32_t* ownord = (uint32_t*)(lds + L.ownord);
        uint32_t* srcL = (uint32_t*)(lds + L.srcL);
        uint32_t* basec = (uint32_t*)(lds + L.basec);
        uint32_t* cq = (uint32_t*)(lds + L.cq);
        uint32_t* co = (uint32_t*)(lds + L.co);
        uint8_t* ctxL = lds + L.ctxL;
        uint8_t* ncand = lds + L.ncand;
        uint16_t* wg = (uint16_t*)(lds + L.wg);
        uint8_t* ownv = lds + L.ownv;
        uint8_t* ownml = lds + L.ownml;
        uint8_t* ownE = lds + L.ownE;
        uint8_t* oldml = lds + L.oldml;
        uint8_t* oldE = lds + L.oldE;
        uint8_t* tyL = lds + L.tyL;
        uint8_t* w0L = lds + L.w0L;
        uint8_t* lrL = lds + L.lrL;
        uint8_t* cnt = lds + L.cnt;
        uint8_t* cml = lds + L.cml;
        uint8_t* cl = lds + L.cl;
        uint32_t* scal = (uint32_t*)(lds + L.scal);  // [0] p, [1] lt, [2] changed

        const uint8_t* b = a.win;
        const uint32_t seg_start = kPre + sg * a.seg;
        const uint32_t seg_end = seg_start + a.seg < a.len ? seg_start + a.seg : a.len;
        const uint32_t npos = seg_end - seg_start;          // item-start positions of the segment
        const uint32_t nprobe = npos + 2;                   // + lazy probe positions
        const uint32_t D = a.dmax;

        // ---- phase 0: stage the segment's bytes, slots, old state and the ctx ordinals in LDS
        for (uint32_t i = lane; i < kLbLen; i += 64) lb[i] = b[(int64_t)seg_start - kLbPre + i];
        for (uint32_t x = lane; x < nprobe; x += 64) {
            const uint32_t pos = seg_start + x;
            uint32_t j = pos < a.len ? a.idx[pos] : 0;
            idxL[x] = j;
            ncand[x] = 0;
            if (x < npos) {
                oldml[x] = a.sml[j];
                uint32_t oe = 0;
                if (pos >= kPre + 1) {
                    uint32_t ks = a.kidx[pos - 2];
                    oe = (uint32_t)((a.kbits[ks >> 6] >> (ks & 63)) & 1);
                }
                oldE[x] = (uint8_t)oe;
                ownv[x] = 0; ownml[x] = 0; ownE[x] = 0;
            }
        }
        for (uint32_t c = lane; c < 256; c += 64) {
            basec[c] = a.base[(size_t)(sg % a.ring) * 256 + c];
            cnt[c] = 0;
        }
        w.sync();
        for (uint32_t x = lane; x < nprobe; x += 64) {
            const uint8_t* px = lb + kLbPre + x;
            uint32_t c = (uint32_t)(px[-1] & 0x7f) | ((uint32_t)is_alnum(px[-2]) << 7);  // hash1(pos-1)
            ctxL[x] = (uint8_t)c;
            keyL[x] = c * kHash + hash_entry(px);
        }
        for (uint32_t i = lane; i < nprobe + 2; i += 64) {  // position u = seg_start - 2 + i: hash2(u - 1)
            const uint8_t* pu = lb + kLbPre - 2 + i;
            uint32_t h1 = (uint32_t)(pu[-2] & 0x7f) | ((uint32_t)is_alnum(pu[-3]) << 7);
            kkL[i] = (uint32_t)(pu[-1] & 0x7f) | (h1 << 7);
        }
        w.sync();

        // ---- phase 1: every position of the segment collects, on its own lane, the candidates that
        // older segments offer it: the most recent <= D ring members of its (ctx, hash) run, their
        // ordinals, expected lengths and common-prefix lengths; and the word predictor's answer.
        for (uint32_t x = lane; x < nprobe; x += 64) {
            const uint32_t pos = seg_start + x;
            if (pos >= a.len) continue;
            const uint32_t key = keyL[x];
            uint32_t hi = idxL[x];
            for (uint32_t y = 0; y < x; y++)
                if (keyL[y] == key) { hi = idxL[y]; break; }  // slots >= hi belong to this segment
            const uint32_t lo = a.runstart[key];
            uint32_t found = 0;
            uint32_t* mycq = cq + x * D;
            if (hi > lo) {
                uint32_t wi = (hi - 1) >> 6;
                uint64_t word = a.vbits[wi];
                if (hi & 63) word &= (1ull << (hi & 63)) - 1;
                for (;;) {
                    const uint32_t wlo = wi << 6;
                    if (wlo < lo) word &= ~0ull << (lo - wlo);
                    while (word && found < D) {
                        int bit = 63 - clz64(word);
                        mycq[found++] = wlo + (uint32_t)bit;
                        word &= ~(1ull << bit);
                    }
                    if (found >= D || wlo <= lo) break;
                    wi--;
                    word = a.vbits[wi];
                }
            }
            uint32_t* myco = co + x * D;
            uint8_t* myml = cml + x * D;
            uint8_t* mycl = cl + x * D;
            const uint8_t* px = lb + kLbPre + x;
            const uint64_t x0 = ldu64(px);
            // eight candidates a round: all slot reads, then all first-8-byte compares, in flight together
            for (uint32_t k0 = 0; k0 < found; k0 += 8) {
                uint32_t sl[8], q[8], m[8], o[8];
                uint64_t d0[8];
#pragma unroll
                for (int i = 0; i < 8; i++) sl[i] = mycq[k0 + i < found ? k0 + i : k0];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    q[i] = a.epos[sl[i]];
                    m[i] = a.sml[sl[i]];
                    o[i] = a.sord[sl[i]];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) d0[i] = ldu64(b + q[i]) ^ x0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (k0 + i >= found) break;
                    uint32_t l;
                    if (d0[i]) l = (uint32_t)ctz64(d0[i]) >> 3;
                    else l = 8 + lcp240u(b + q[i] + 8, px + 8, kMaxLen - 8);
                    myco[k0 + i] = o[i];
                    myml[k0 + i] = (uint8_t)m[i];
                    mycq[k0 + i] = q[i];
                    mycl[k0 + i] = m[i] == 255 ? 0 : (uint8_t)l;
                }
            }
            ncand[x] = (uint8_t)found;
            if (x < npos) {  // words[hash2(pos-1)] as older segments leave it
                const uint32_t kk = kkL[x + 2];
                uint32_t khi = a.kidx[pos];
                for (uint32_t i = 0; i < x + 2; i++) {
                    const uint32_t u = seg_start - 2 + i;
                    if (u >= kPre - 1 && kkL[i] == kk) { khi = a.kidx[u]; break; }
                }
                const uint32_t klo = a.krun[kk];
                uint32_t wv = (uint32_t)a.wsnap[kk * 2] | ((uint32_t)a.wsnap[kk * 2 + 1] << 8);
                if (khi > klo) {
                    uint32_t wi = (khi - 1) >> 6;
                    uint64_t word = a.kbits[wi];
                    if (khi & 63) word &= (1ull << (khi & 63)) - 1;
                    for (;;) {
                        const uint32_t wlo = wi << 6;
                        if (wlo < klo) word &= ~0ull << (klo - wlo);
                        if (word) {
                            const uint32_t u = a.kpos[wlo + (uint32_t)(63 - clz64(word))];
                            wv = (uint32_t)b[u] | ((uint32_t)b[u + 1] << 8);
                            break;
                        }
                        if (wlo <= klo) break;
                        wi--;
                        word = a.kbits[wi];
                    }
                }
                wg[x] = (uint16_t)wv;
            }
        }
        // ---- entry state: where the previous segment left the stream (look through skipped ones)
        if (lane == 0) {
            uint32_t p, lt;
            if (sg == 0) {
                p = kPre;
                lt = a.lt0;
            } else {
                uint32_t v = a.exitst[sg];
                for (uint32_t d = 1; d <= 4 && d < sg; d++) {
                    uint32_t v2 = a.exitst[sg - d];
                    if (v2 > v) v = v2;
                }
                p = v >> 2;
                lt = v & 3;
                if (p < seg_end) ownE[p - seg_start] = (lt != kTyWord);
            }
            scal[0] = p; scal[1] = lt; scal[2] = 0;
        }
        w.sync();

        // ---- phase 2: the segment's items, one after another, with the reference's decision rules
        for (;;) {
            const uint32_t p = scal[0];
            if (p >= seg_end) break;
            const uint32_t x = p - seg_start;
            // this sweep's own earlier items are the most recent candidates (and word updates)
            const bool ownc = lane < x && ownv[lane];
            const uint32_t kl = lane < npos ? keyL[lane] : 0xffffffffu;
            const uint64_t m0 = w.ballot(ownc && kl == keyL[x]);
            const uint64_t m1 = w.ballot(ownc && kl == keyL[x + 1]);
            const uint64_t m2 = w.ballot(ownc && kl == keyL[x + 2]);
            const uint64_t mE = w.ballot(lane <= x && lane < npos && ownE[lane] && kkL[lane] == kkL[x + 2]);
            if (lane == 0) {
                const uint8_t* px = lb + kLbPre + x;
                uint32_t lt = scal[1];
                const uint32_t c = ctxL[x];
                uint32_t w0, w1;
                if (mE) {
                    const uint32_t y = 63 - (uint32_t)clz64(mE);
                    w0 = lb[kLbPre + y - 2];
                    w1 = lb[kLbPre + y - 1];
                } else {
                    w0 = wg[x] & 0xff;
                    w1 = wg[x] >> 8;
                }
                const uint32_t lwm = (px[0] == w0 && px[1] == w1);
                // find_match, src/matcher.rs:135-192
                const uint32_t hcnt = basec[c] + cnt[c];
                uint32_t max_len = kMinLen - 1, mlexp = kMinLen, bestq = 0, besto = 0, cntv = 0;
                bool stop = false;
                for (uint64_t m = m0; m && !stop;) {
                    const uint32_t y = 63 - (uint32_t)clz64(m);
                    m &= ~(1ull << y);
                    const uint32_t oq = ownord[y];
                    if (hcnt - 1 - oq > kRing - 1 || cntv >= a.depth) { stop = true; break; }
                    cntv++;
                    const uint32_t l = lcp240u(lb + kLbPre + y, px);
                    if (l > max_len) {
                        mlexp = ownml[y]; max_len = l; bestq = seg_start + y; besto = oq;
                        if (l == kMaxLen || (mlexp > 0 && l > mlexp)) stop = true;
                    } else if (l + 3 < max_len && mlexp > 0 && l > mlexp) {
                        if (ldu32(lb + kLbPre + y + max_len - 3) == ldu32(px + max_len - 3)) stop = true;
                    }
                }
                const uint32_t nc = ncand[x];
                for (uint32_t k = 0; k < nc && !stop; k++) {
                    const uint32_t ml = cml[x * D + k];
                    if (ml == 255) continue;
                    const uint32_t oq = co[x * D + k];
                    if (hcnt - 1 - oq > kRing - 1 || cntv >= a.depth) break;
                    cntv++;
                    const uint32_t l = cl[x * D + k];
                    if (l > max_len) {
                        mlexp = ml; max_len = l; bestq = cq[x * D + k]; besto = oq;
                        if (l == kMaxLen || (mlexp > 0 && l > mlexp)) break;
                    } else if (l + 3 < max_len && mlexp > 0 && l > mlexp) {
                        // the reference's 4-byte prefilter can pass by chance past the mismatch; it then
                        // leaves the walk without a better match (src/matcher.rs:150-168)
                        if (ldu32(b + cq[x * D + k] + max_len - 3) == ldu32(px + max_len - 3)) break;
                    }
                }
                const bool is_match = max_len >= kMinLen && p + max_len < a.len;
                uint32_t lazy = 0;
                if (is_match && max_len < kMaxLen / 2) {  // src/lz.rs:151-170
                    const uint32_t ro = hcnt - 1 - besto;
                    const uint32_t l1 = max_len + 1 + (roid_bitlen(ro) < 8), l2 = l1 - lwm;
                    if (has_lazy(lds, L, D, m1, x + 1, l1, a.lazy1)) lazy = 1;
                    else if (has_lazy(lds, L, D, m2, x + 2, l2, a.lazy2)) lazy = 2;
                }
                // commit, src/lz.rs:172-234
                ownv[x] = 1;
                ownord[x] = hcnt;
                lrL[x] = cnt[c];
                cnt[c]++;
                w0L[x] = (uint8_t)w0;
                const uint8_t al = (lt == kTyLit) ? 4 : 0;
                uint32_t np;
                if (is_match && !lazy) {
                    ownml[x] = (uint8_t)max_len; srcL[x] = bestq; tyL[x] = kTyMatch | al;
                    np = p + max_len; lt = kTyMatch;
                } else if (p + 1 < a.len && lazy != 1 && lwm) {
                    ownml[x] = 0; tyL[x] = kTyWord | al;
                    np = p + 2; lt = kTyWord;
                } else {
                    ownml[x] = 0; tyL[x] = kTyLit | al;
                    np = p + 1; lt = kTyLit;
                }
                if (np < seg_end) ownE[np - seg_start] = (lt != kTyWord);
                scal[0] = np; scal[1] = lt;
            }
            w.sync();
        }

        // ---- phase 3: publish what changed, the per-ctx item counts and the exit state
        bool changed = false;
        for (uint32_t x = lane; x < npos; x += 64) {
            const uint32_t pos = seg_start + x;
            const uint32_t j = idxL[x];
            const uint32_t nm = ownv[x] ? ownml[x] : 255u;
            const uint32_t om = oldml[x];
            if (nm != om) {
                a.sml[j] = (uint8_t)nm;
                if ((nm == 255) != (om == 255)) {
                    if (nm == 255) atom_and64(&a.vbits[j >> 6], ~(1ull << (j & 63)));
                    else atom_or64(&a.vbits[j >> 6], 1ull << (j & 63));
                }
                changed = true;
            }
            if (ownv[x]) {
                a.TY[pos] = tyL[x];
                a.W0[pos] = w0L[x];
                a.LR[pos] = lrL[x];
                if ((tyL[x] & 3) == kTyMatch) a.SRC[pos] = srcL[x];
            }
            if (pos >= kPre + 1 && ownE[x] != oldE[x]) {
                const uint32_t ks = a.kidx[pos - 2];
                if (ownE[x]) atom_or64(&a.kbits[ks >> 6], 1ull << (ks & 63));
                else atom_and64(&a.kbits[ks >> 6], ~(1ull << (ks & 63)));
                changed = true;
            }
        }
        {
            uint32_t* part = a.partial + ((size_t)a.par * (a.wsegs / kRankChunk + 1) + w.block() / kRankChunk) * 256;
            for (uint32_t c = lane; c < 256; c += 64) {
                a.hist[(size_t)(sg % a.ring) * 256 + c] = cnt[c];
                if (cnt[c]) atom_add32(&part[c], cnt[c]);
            }
        }
        if (lane == 0) {
            const uint32_t v = (scal[0] << 2) | scal[1];
            if (a.exitst[sg + 1] != v) {
                a.exitst[sg + 1] = v;
                changed = true;
            }
            atom_add32(&a.ctl->evals, 1);
        }
        if (w.ballot(changed) && lane == 0) atom_min32(&a.ctl->fchg[a.par], sg);
    }

    // has_lazy_match (src/matcher.rs:194-228) for probe position xx in {x+1, x+2}: candidates are the
    // ring members inserted before p, newest first: this sweep's own items (mask), then the older
    // segments' list collected in phase 1.
    ORZ_D static bool has_lazy(uint8_t* lds, const ParseLds& L, uint32_t D, uint64_t mown, uint32_t xx, uint32_t min_len,
                               uint32_t depth) {
        const uint8_t* lb = lds + L.lb;
        const uint32_t* ownord = (const uint32_t*)(lds + L.ownord);
        const uint32_t* basec = (const uint32_t*)(lds + L.basec);
        const uint8_t* cnt = lds + L.cnt;
        const uint8_t* ctxL = lds + L.ctxL;
        const uint32_t cx = ctxL[xx];
        const uint32_t hx = basec[cx] + cnt[cx];
        uint32_t cntv = 0;
        for (uint64_t m = mown; m;) {
            const uint32_t y = 63 - (uint32_t)clz64(m);
            m &= ~(1ull << y);
            if (hx - 1 - ownord[y] > kRing - 1 || cntv >= depth) return false;
            cntv++;
            if (lcp240u(lb + kLbPre + y, lb + kLbPre + xx) >= min_len) return true;
        }
        const uint32_t* co = (const uint32_t*)(lds + L.co);
        const uint8_t* cml = lds + L.cml;
        const uint8_t* cl = lds + L.cl;
        const uint32_t nc = (lds + L.ncand)[xx];
        for (uint32_t k = 0; k < nc; k++) {
            if (cml[xx * D + k] == 255) continue;
            if (hx - 1 - co[xx * D + k] > kRing - 1 || cntv >= depth) return false;
            cntv++;
            if (cl[xx * D + k] >= min_len) return true;
        }
        return false;
    }
};

// ---------------------------------------------------------------------------------------------
// After a sweep: ring ordinals.  One block of 256 threads (thread = ctx c) per chunk of
// kRankChunk segments of the window:
//   off[c]       = base[front][c] + sum of the partial[chunk' < chunk][c] that ParseWave accumulated
//   base[s+1][c] = base[s][c] + hist[s][c] over the chunk's segments            (ring of R rows)
//   sord         = base[seg][ctx] + LR for every current item of the chunk      (what RankApply was)
// Block 0 also moves the front to just past the first changed segment and re-arms the other parity.
struct RankArgs {
    const uint8_t* win;
    ParseCtl* ctl;
    const uint8_t* hist;
    uint32_t* base;
    uint32_t* partial;
    const uint32_t* idx;
    const uint8_t* sml;
    const uint8_t* LR;
    uint32_t* sord;
    uint32_t nseg, seg, wsegs, ring, len, par;
};
// `rows` = LDS [kRankChunk + 1][256] u32 ; sync() = block barrier ; c = thread id (0..255)
template <class SYNC>
ORZ_D void rank_chunk(const RankArgs& a, uint32_t chunk, uint32_t c, uint32_t* rows, SYNC sync) {
    const uint32_t f = a.ctl->front[a.par];
    const uint32_t wend = f + a.wsegs < a.nseg ? f + a.wsegs : a.nseg;
    const uint32_t nchunk = a.wsegs / kRankChunk + 1;
    const uint32_t s0 = f + chunk * kRankChunk;
    const uint32_t s1 = s0 + kRankChunk < wend ? s0 + kRankChunk : wend;
    uint32_t* part = a.partial + (size_t)a.par * nchunk * 256;
    if (s0 < wend) {
        uint32_t run = a.base[(size_t)(f % a.ring) * 256 + c];
        for (uint32_t i = 0; i < chunk; i++) run += part[i * 256 + c];
        rows[c] = run;
        for (uint32_t s = s0; s < s1; s++) {
            run += a.hist[(size_t)(s % a.ring) * 256 + c];
            rows[(s - s0 + 1) * 256 + c] = run;
            a.base[(size_t)((s + 1) % a.ring) * 256 + c] = run;
        }
    }
    sync();
    if (s0 < wend) {
        const uint32_t x0 = kPre + s0 * a.seg;
        const uint32_t npos = (s1 - s0) * a.seg;
        for (uint32_t i = c; i < npos; i += 256) {
            const uint32_t x = x0 + i;
            if (x >= a.len) break;
            const uint32_t j = a.idx[x];
            if (a.sml[j] != 255) a.sord[j] = rows[(i / a.seg) * 256 + hash1(a.win, x - 1)] + a.LR[x];
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
    }
}

}  // namespace orz
