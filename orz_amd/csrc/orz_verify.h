// orz_verify.h -- the validity gate of the encoder: every block's ITEMS, as they are about to be coded, against the
// DECODER's rules (LZDecoder::decode, /root/reference/src/lz.rs:366-478), on the device, before a byte of the block is
// handed out.  A finding fails the encode: no stream is better than one no decoder follows.
//
// The gate shares nothing with the parse it checks.  It reads the item arrays of the post stage (position, symbol,
// context, excluded symbol, length code, offset bits: what ItemBits / Pack will write) plus SRC / ML per position,
// and carries the decoder's state on its own, block to block:
//   vrec[x]    per window offset: is an item start | its ring context | its match length | the len_min its ring node has
//              reached (src/matcher.rs:65-71), kept by the gate itself                         (slides with the window)
//   vord[x]    ordinal of the item start x in its context's ring, counted by the gate itself   (slides with the window)
//   vctx[256]  items each ring has taken so far            (src/matcher.rs:62-80: head arithmetic)
//   vwords     the words[] table                           (src/lz.rs:132-133,203,233)
//   vlast      was the last item a literal                 (after_literal, src/lz.rs:66)
// Neither the parse's bitmaps, its ordinals (ORD), its word-update lists, its len_min values (LMV / LENMIN: since round 5 --
// until then the gate read the very value ItemSyms codes from) nor the contexts it derived are consulted: the
// context of a source is the one RECORDED when the source was an item (round 3's slide defect lived in re-deriving it
// from the window's bytes, and FastVerify, which re-derived it the same way, could not see it); ordinals come from the
// stable order of the item list (the two after_literal runs of a context merged by position); words[] answers come from
// the sorted list of the items' own update events.
//
// Classes of findings (VerErr): a hole or overlap in the item sequence (the round-3 defect: an item rewritten without its
// span being re-parsed), after_literal / context / symbol not what the decoder will compute, a source that is no item
// start / of another ring / outside the ring / with other bytes, an offset code that is not the ring distance, a length
// below the source's len_min or a length code the decoder reads as another length (src/lz.rs:459-467), an excluded symbol
// or a WORD that the words[] table does not give.
// What it does not cover: symbol ranking, Huffman tables and bit packing (downstream of the items) -- those are covered
// by the ranking guard (backend symrank) and, on request, by a full decode of the finished stream (ORZ_VERIFY=decode).
#pragma once
#include "orz_kernels.h"

namespace orz {

enum VerErr : uint32_t {
    kVeTiling = 0,   // item k does not end where item k + 1 starts (or the first / last item misses the block's ends)
    kVeAfterLit,     // after_literal flag != "the previous item was a literal"
    kVeContext,      // coding context != hash1 of the bytes before the item | after_literal << 8
    kVeSymbol,       // literal symbol != the byte; match flag / length range inconsistent
    kVeSource,       // source position not below the item, or no item start (by the gate's own record)
    kVeSourceCtx,    // source recorded under another ring context
    kVeBytes,        // source bytes != item bytes
    kVeRing,         // ring distance (gate's ordinals) > 4093
    kVeOffsetCode,   // symbol / offset bits do not encode that ring distance
    kVeOrdinal,      // the parse's ordinal (ORD) != the gate's count
    kVeLenMin,       // length < max(len_min of the source at that time, 4): not representable
    kVeLenCode,      // the decoder would read another length from (enc, len_min, len_expected)
    kVeUnlikely,     // excluded symbol != words[hash2][0]
    kVeWord,         // WORD item whose two bytes the table does not predict
    kVeFirst,        // position of the first finding + 1 (any class)
    kVeCount
};
ORZ_HD const char* ver_name(uint32_t e) {
    const char* n[] = {"hole/overlap in the item sequence", "after_literal", "context", "symbol", "source is no item start", "source in another ring",
                       "source bytes differ", "source outside the ring", "offset code", "ordinal", "length below len_min", "length code",
                       "excluded symbol", "WORD prediction"};
    return e < kVeFirst ? n[e] : "?";
}

constexpr uint32_t kVrValid = 1u << 31;
ORZ_HD uint32_t vrec_make(uint32_t ctx8, uint32_t mlen) { return kVrValid | (ctx8 << 16) | (mlen << 8); }  // (len_min 0: a new node)
ORZ_HD uint32_t vrec_ctx(uint32_t r) { return (r >> 16) & 0xff; }
ORZ_HD uint32_t vrec_mlen(uint32_t r) { return (r >> 8) & 0xff; }
ORZ_HD uint32_t vrec_lenmin(uint32_t r) { return r & 0x7f; }

struct VerArgs {
    const uint8_t* win;
    const uint32_t* ipos;
    const uint16_t *isym, *ictx, *irob;
    const uint8_t *iunl, *ienc, *ial;
    uint32_t nitems;
    uint32_t end;  // window offset where the block's last item must end
    const uint8_t* ML;
    const uint32_t *SRC, *ORD;
    const uint32_t *sperm, *rstart;  // items in (context | after_literal << 8) order, stable; run starts [513]
    uint32_t *vrec, *vord, *vctx, *vlast;
    uint8_t* vwords;
    uint32_t* err;  // [kVeCount]
};
ORZ_HD void ver_fail(const VerArgs& a, uint32_t cls, uint32_t p) {
    ORZ_ATOMIC_ADD(&a.err[cls], 1u);
    ORZ_ATOMIC_MIN(&a.err[kVeFirst], p + 1);
}
ORZ_HD uint32_t ver_len(const VerArgs& a, uint32_t k) {
    const uint32_t s = a.isym[k];
    return s < 256 ? 1u : (s == kWordSym ? 2u : (uint32_t)a.ML[a.ipos[k]]);
}

struct VerInit {  // err[first] starts at "none"
    uint32_t* err;
    ORZ_HD void operator()(size_t t) const {
        if (t < kVeCount) err[t] = t == kVeFirst ? ~0u : 0u;
    }
};
// thread per item: the sequence, the flags and contexts the decoder will compute, the records of this block's item starts
struct VerItems {
    VerArgs a;
    ORZ_HD void operator()(size_t k) const {
        if (k >= a.nitems) return;
        const uint32_t p = a.ipos[k], s = a.isym[k], al = a.ial[k] & 1, ism = (a.ial[k] >> 1) & 1;
        const uint32_t len = ver_len(a, (uint32_t)k);
        const uint32_t next = k + 1 < a.nitems ? a.ipos[k + 1] : a.end;
        if ((k == 0 && p != kPre) || p + len != next) ver_fail(a, kVeTiling, p);
        const uint32_t prev_lit = k ? (uint32_t)(a.isym[k - 1] < 256) : a.vlast[0];
        if (al != prev_lit) ver_fail(a, kVeAfterLit, p);
        const uint32_t c = hash1(a.win, p - 1);
        if (a.ictx[k] != (c | (al << 8))) ver_fail(a, kVeContext, p);
        const bool match = s >= 256 && s != kWordSym;
        if (s >= kSyms || (s < 256 && s != a.win[p]) || (match != (ism != 0)) || (match && (len < kMinLen || len > kMaxLen))) ver_fail(a, kVeSymbol, p);
        a.vrec[p] = vrec_make(c, match ? len : 0);
        if (k + 1 == a.nitems) a.vlast[1] = s < 256;
    }
};
// thread per slot of the context-sorted item list: the gate's own ring ordinals.  The list is sorted, stably, by
// (context | after_literal << 8): the items of a ring are two runs in position order, an item's ordinal is its rank in
// its own run plus the items of the sister run before it.
struct VerOrdinals {
    VerArgs a;
    ORZ_HD void operator()(size_t j) const {
        if (j >= a.nitems) return;
        const uint32_t k = a.sperm[j], key = a.ictx[k] & 511, sis = key ^ 256;
        const uint32_t own = (uint32_t)j - a.rstart[key];
        uint32_t lo = a.rstart[sis], hi = a.rstart[sis + 1];
        const uint32_t s0 = lo;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (a.sperm[mid] < k) lo = mid + 1; else hi = mid;
        }
        const uint32_t p = a.ipos[k];
        const uint32_t ord = a.vctx[key & 255] + own + (lo - s0);
        a.vord[p] = ord;
        if (a.ORD[p] != ord) ver_fail(a, kVeOrdinal, p);
    }
};
// thread per item, matches only: the source by the gate's records, the ring distance by the gate's ordinals, the codes by
// the decoder's formulas
struct VerMatches {
    VerArgs a;
    ORZ_HD void operator()(size_t k) const {
        if (k >= a.nitems) return;
        const uint32_t s = a.isym[k];
        if (!((a.ial[k] >> 1) & 1)) return;
        const uint32_t p = a.ipos[k], L = a.ML[p], q = a.SRC[p];
        if (q < 1 || q >= p) { ver_fail(a, kVeSource, p); return; }
        const uint32_t rec = a.vrec[q];
        if (!(rec & kVrValid)) { ver_fail(a, kVeSource, p); return; }
        if (vrec_ctx(rec) != (uint32_t)(a.ictx[k] & 255)) ver_fail(a, kVeSourceCtx, p);
        bool same = true;
        for (uint32_t i = 0; i < L && same; i++) same = a.win[q + i] == a.win[p + i];
        if (!same) ver_fail(a, kVeBytes, p);
        const uint32_t ro = a.vord[p] - 1 - a.vord[q];
        if (ro > kRing - 1) { ver_fail(a, kVeRing, p); return; }
        uint32_t roid, bl, bits;
        roid_encode(ro, &roid, &bl, &bits);
        const uint32_t enc = a.ienc[k];
        if (s != 256 + roid * 6 + (enc < 5 ? enc : 5) || a.irob[k] != (bits | (bl << 12))) ver_fail(a, kVeOffsetCode, p);
        // (len_min and the length code: VerLenMin below, with the gate's own len_min)
    }
};
// len_min (src/matcher.rs:65-71): a ring node starts at 0 and every reference raises it to min(len + 1, 127); a reference must
// be at least max(len_min, 4) long and its length code is read against it (src/lz.rs:459-467).  The gate's own bookkeeping: the
// references of the block sorted by (source, item index) -- keys and sort of its own --, the value a reference meets = the
// node's value from earlier blocks (in vrec) and the references before it in its source's run.  A walk back stops at 127 (the
// cap) -- and after 256 references below 126, which cannot all have been longer than the one before: a finding by itself.
struct VerLmKeys {  // thread per item: source << 25 | item index for matches, ~0 for the others
    VerArgs a;
    uint64_t* keys;
    ORZ_HD void operator()(size_t k) const {
        if (k >= a.nitems) return;
        keys[k] = ((a.ial[k] >> 1) & 1) ? (((uint64_t)a.SRC[a.ipos[k]] << kPosBits) | (uint64_t)k) : ~0ull;
    }
};
ORZ_HD uint32_t ver_lm_before(const VerArgs& a, const uint64_t* keys, size_t j, uint32_t q, bool* runaway) {
    uint32_t v = vrec_lenmin(a.vrec[q]);
    uint32_t steps = 0;
    for (size_t i = j; i > 0 && v < 127;) {
        i--;
        const uint64_t ki = keys[i];
        if ((uint32_t)(ki >> kPosBits) != q) break;
        if (++steps > 256) { *runaway = true; break; }
        const uint32_t l = a.ML[a.ipos[(uint32_t)(ki & kPosMask)]];
        const uint32_t w = l + 1 < 127 ? l + 1 : 127;
        if (w > v) v = w;
    }
    return v;
}
struct VerLenMin {  // thread per sorted reference
    VerArgs a;
    const uint64_t* keys;
    ORZ_HD void operator()(size_t j) const {
        if (j >= a.nitems) return;
        const uint64_t key = keys[j];
        if (key == ~0ull) return;
        const uint32_t q = (uint32_t)(key >> kPosBits), k = (uint32_t)(key & kPosMask), p = a.ipos[k], L = a.ML[p];
        if (q < 1 || q >= p || !(a.vrec[q] & kVrValid)) return;  // (reported by VerMatches)
        bool runaway = false;
        const uint32_t lm = ver_lm_before(a, keys, j, q, &runaway);
        if (runaway) { ver_fail(a, kVeLenMin, p); return; }
        const uint32_t m = lm > kMinLen ? lm : kMinLen;
        const uint32_t rec = a.vrec[q], e = vrec_mlen(rec) > kMinLen ? vrec_mlen(rec) : kMinLen;
        if (L < m) ver_fail(a, kVeLenMin, p);
        const uint32_t enc = a.ienc[k];
        const uint32_t dec = enc + m > e ? enc + m : (enc > 0 ? enc + m - 1 : e);  // src/lz.rs:459-467
        if (dec != L || enc > kLenSyms - 1) ver_fail(a, kVeLenCode, p);
    }
};
struct VerLmCommit {  // the last reference of each source's run leaves the node's new len_min in the gate's record (a launch of its own:
    VerArgs a;        // the references of the run read the old one)
    const uint64_t* keys;
    ORZ_HD void operator()(size_t j) const {
        if (j >= a.nitems) return;
        const uint64_t key = keys[j];
        if (key == ~0ull) return;
        const uint32_t q = (uint32_t)(key >> kPosBits);
        if (j + 1 < a.nitems && keys[j + 1] != ~0ull && (uint32_t)(keys[j + 1] >> kPosBits) == q) return;
        const uint32_t k = (uint32_t)(key & kPosMask), p = a.ipos[k];
        if (q < 1 || q >= p || !(a.vrec[q] & kVrValid)) return;
        bool runaway = false;
        uint32_t v = ver_lm_before(a, keys, j, q, &runaway);
        const uint32_t l = a.ML[p], w = l + 1 < 127 ? l + 1 : 127;
        if (w > v) v = w;
        if (runaway) v = 127;
        a.vrec[q] = (a.vrec[q] & ~0x7fu) | v;
    }
};
// words[] (src/lz.rs:132-133,203,233): an item that is not a WORD, ending at y, writes words[hash2(y - 3)] = the two bytes
// before y; an item starting at p reads words[hash2(p - 1)] after every update of items ending at or before p.  Events
// (key << 25 | y) are sorted; a read is a search for the last event of its key not after p.
struct VerWordEvents {
    VerArgs a;
    uint64_t* ev;
    ORZ_HD void operator()(size_t k) const {
        if (k >= a.nitems) return;
        const uint32_t y = a.ipos[k] + ver_len(a, (uint32_t)k);
        ev[k] = a.isym[k] == kWordSym ? ~0ull : (((uint64_t)hash2(a.win, y - 3) << kPosBits) | y);
    }
};
struct VerWords {
    VerArgs a;
    const uint64_t* ev;  // sorted
    ORZ_HD void operator()(size_t k) const {
        if (k >= a.nitems) return;
        const uint32_t p = a.ipos[k], key = hash2(a.win, p - 1);
        const uint64_t target = ((uint64_t)key << kPosBits) | p;
        uint32_t lo = 0, hi = a.nitems;  // first event > target
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (ev[mid] <= target) lo = mid + 1; else hi = mid;
        }
        uint32_t w0, w1;
        if (lo > 0 && (uint32_t)(ev[lo - 1] >> kPosBits) == key) {
            const uint32_t y = (uint32_t)(ev[lo - 1] & kPosMask);
            w0 = a.win[y - 2]; w1 = a.win[y - 1];
        } else {
            w0 = a.vwords[key * 2]; w1 = a.vwords[key * 2 + 1];
        }
        if (a.iunl[k] != w0) ver_fail(a, kVeUnlikely, p);
        if (a.isym[k] == kWordSym && (a.win[p] != w0 || a.win[p + 1] != w1)) ver_fail(a, kVeWord, p);
    }
};
struct VerWordsCarry {  // the last event of each key becomes the table entry the next block starts from
    VerArgs a;
    const uint64_t* ev;
    ORZ_HD void operator()(size_t j) const {
        if (j >= a.nitems || ev[j] == ~0ull) return;
        const uint32_t key = (uint32_t)(ev[j] >> kPosBits);
        if (j + 1 < a.nitems && ev[j + 1] != ~0ull && (uint32_t)(ev[j + 1] >> kPosBits) == key) return;
        const uint32_t y = (uint32_t)(ev[j] & kPosMask);
        a.vwords[key * 2] = a.win[y - 2];
        a.vwords[key * 2 + 1] = a.win[y - 1];
    }
};
struct VerCarry {  // ring counts and after_literal for the next block
    VerArgs a;
    ORZ_HD void operator()(size_t c) const {
        if (c >= 256) return;
        a.vctx[c] += (a.rstart[c + 1] - a.rstart[c]) + (a.rstart[c + 257] - a.rstart[c + 256]);
        if (c == 0 && a.nitems) a.vlast[0] = a.vlast[1];
    }
};
// The gate in three launches (round 6; before: nine).  What must be ordered stays ordered -- the records and ordinals of ALL items
// (stage 1) before any match is judged against them (stage 2), every reference judged against the nodes' OLD len_min and the
// table's OLD entries (stage 2) before the new ones are left behind (stage 3) -- and the kernels between those points, which read
// the same item arrays over and over, run as one grid each.  A thread is an item (stages 1, 2) / a slot of the sorted lists (stage 3).
struct VerStage1 {  // records + ordinals + the keys of both sorts
    VerArgs a;
    uint64_t *lmkeys, *wev;
    ORZ_HD void operator()(size_t k) const {
        VerItems{a}(k);
        VerOrdinals{a}(k);
        VerLmKeys{a, lmkeys}(k);
        VerWordEvents{a, wev}(k);
    }
};
struct VerStage2 {  // sources, ring distances and offset codes; len_min and length codes; excluded symbols and WORD items
    VerArgs a;
    const uint64_t *lmk, *evs;  // sorted
    ORZ_HD void operator()(size_t k) const {
        VerMatches{a}(k);
        VerLenMin{a, lmk}(k);
        VerWords{a, evs}(k);
    }
};
struct VerStage3 {  // the state the next block's gate starts from
    VerArgs a;
    const uint64_t *lmk, *evs;
    ORZ_HD size_t threads() const { return a.nitems > 256 ? a.nitems : 256; }
    ORZ_HD void operator()(size_t j) const {
        VerLmCommit{a, lmk}(j);
        VerWordsCarry{a, evs}(j);
        VerCarry{a}(j);
    }
};
struct VerReset {  // LZContext::new (src/lz.rs:57-66): empty rings, zero words[], after_literal = true
    uint32_t *vctx, *vlast;
    ORZ_HD void operator()(size_t c) const {
        if (c < 256) vctx[c] = 0;
        if (c == 0) { vlast[0] = 1; vlast[1] = 1; }
    }
};

// Fault injection for the gate's tests (ORZ_VERIFY_INJECT=<class>:<n>): the n-th suitable item of every block is damaged
// AFTER the parse and BEFORE the items are built, the way a parse defect would.  Thread 0 only.
enum VerInject : uint32_t { kViNone = 0, kViHole, kViContext, kViRing, kViLenMin, kViWord, kViBytes, kViLenMin2 };
struct VerInjectLmv {  // "lenmin2": the len_min the parse hands to ItemSyms damaged AFTER LenMinEval / LenMinCommit -- what the gate could not
    uint32_t nth;      // see while it read that very value: the n-th match coded against a len_min above 4
    const uint32_t* ipos;
    uint32_t nitems;
    const uint8_t* TY;
    uint8_t* LMV;
    ORZ_HD void operator()(size_t t) const {
        if (t) return;
        uint32_t seen = 0;
        for (uint32_t k = 1; k + 1 < nitems; k++) {
            const uint32_t p = ipos[k];
            if ((TY[p] & 3) == kTyMatch && LMV[p] > kMinLen + 1 && seen++ == nth) { LMV[p]--; return; }
        }
    }
};
struct VerInjectK {
    uint32_t kind, nth;
    const uint32_t* ipos;
    uint32_t nitems;
    uint8_t *TY, *ML;
    uint32_t *SRC, *ORD;
    const uint8_t* win;
    const uint8_t* S;
    ORZ_HD void operator()(size_t t) const {
        if (t) return;
        uint32_t seen = 0;
        for (uint32_t k = 1; k + 1 < nitems; k++) {
            const uint32_t p = ipos[k], ty = TY[p] & 3;
            if (kind == kViHole && ty == kTyMatch && ML[p] >= 8) {  // the round-3 defect: a match rewritten as a literal, its span not re-parsed
                if (seen++ == nth) { TY[p] = (uint8_t)((TY[p] & ~3u) | kTyLit); ML[p] = 0; return; }
            } else if (kind == kViContext && ty == kTyMatch) {      // a source of another ring: the item start right after the real source
                const uint32_t q = SRC[p];
                uint32_t q2 = q + 1;
                while (q2 < p && !S[q2]) q2++;
                if (q2 < p && hash1(win, q2 - 1) != hash1(win, p - 1) && seen++ == nth) { SRC[p] = q2; return; }
            } else if (kind == kViRing && ty == kTyMatch) {         // a source more than a ring's length of item starts back in the same context
                if (ORD[SRC[p]] < 2 * kRing) continue;  // (a context with enough history: the walk below is bounded all the same)
                const uint32_t c = hash1(win, p - 1);
                uint32_t q2 = SRC[p], passed = 0, steps = 0;
                while (q2 > 1 && passed < kRing + 4 && steps++ < (1u << 22)) { q2--; if (S[q2] && hash1(win, q2 - 1) == c) passed++; }
                if (passed == kRing + 4 && seen++ == nth) { SRC[p] = q2; return; }
            } else if (kind == kViLenMin && ty == kTyMatch && ML[p] >= 6) {  // the next match of the same ring that is no longer is given the same source:
                const uint32_t c = hash1(win, p - 1);                       // the source's len_min (ML[p] + 1 by then) is above its length
                for (uint32_t k2 = k + 1; k2 + 1 < nitems && k2 < k + 4000; k2++) {
                    const uint32_t p2 = ipos[k2];
                    if ((TY[p2] & 3) == kTyMatch && ML[p2] <= ML[p] && hash1(win, p2 - 1) == c && SRC[p2] != SRC[p]) {
                        if (seen++ == nth) { SRC[p2] = SRC[p]; return; }
                        break;
                    }
                }
            } else if (kind == kViWord && ty == kTyLit && (TY[ipos[k + 1]] & 3) == kTyLit && ipos[k + 1] == p + 1) {
                if (seen++ == nth) { TY[p] = (uint8_t)((TY[p] & ~3u) | kTyWord); return; }  // (two literals called a WORD: the table will not predict them, as a rule)
            } else if (kind == kViBytes && ty == kTyMatch && ML[p] < kMaxLen && win[SRC[p] + ML[p]] != win[p + ML[p]]) {
                if (seen++ == nth) { ML[p]++; return; }             // one byte longer than the common prefix
            }
        }
    }
};

}  // namespace orz
