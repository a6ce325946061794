// orz_stream.h -- host-side orchestration of one orz stream on one device.
//
// StreamEncoder<BE> is the device-side LZEncoder (reference: /root/reference/src/lz.rs:69-346):
// it owns the window, the ring / symrank / word-predictor model state and produces, block by
// block, the framed chunks `orz::encode` (src/lib.rs:58-92) would write.  BE is a backend:
//   HipBackend  (backend_hip.h)  -- the product: HIP kernels on gfx950
//   EmuBackend  (tests/emu)      -- host emulation of the same kernel bodies, tests only
//
// Per block:  prep   (candidate lists: radix sort by (ctx, hash) / by hash2, slot state)
//             parse  fast mode (default): pipelined Gauss-Seidel rounds over tiles, replayed as a hipGraph, then
//                     source assignment + repairs with the item boundaries frozen (orz_fast.h, fast_parse below);
//                     exact mode: speculative sweeps of ParseWave + rank kernel until the front reaches the end
//                     of the block (orz_parse.h)
//             post   (items -> len_min -> symbols on the main stream; symbol ranking on stream 1, launch after
//                     launch; histograms -> Huffman -> bit pack on stream 2 -- post_stage below)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "orz_fast.h"
#include "orz_kernels.h"
#include "orz_parse.h"
#include "orz_verify.h"

namespace orz {

// optional per-item trace of the encoder (parity tests compare it with the oracle's trace)
struct ItemTrace {
    std::vector<uint32_t> block, pos, src;
    std::vector<uint16_t> sym, ctx, rank, rob;
    std::vector<uint8_t> unl, enc, al, mlen;
    void clear() { block.clear(); pos.clear(); src.clear(); sym.clear(); ctx.clear(); rank.clear(); rob.clear(); unl.clear(); enc.clear(); al.clear(); mlen.clear(); }
};

struct EncodeStats {
    uint64_t blocks = 0, sweeps = 0, seg_evals = 0, items = 0, chunks = 0, in_bytes = 0, out_bytes = 0;
    uint64_t rank_redos = 0;  // blocks whose symbol ranking was repeated by the guard (backend symrank)
    uint64_t host_syncs = 0;  // times the host waited for a stream (HIP backend: every hipStreamSynchronize of the encoder's own code)
    double t_prep = 0, t_parse = 0, t_post = 0;  // seconds (host clock around device syncs)
};

// ---- prep kernels (thread per element) -------------------------------------------------------
struct HistFlags32 {
    const uint8_t* S;
    uint32_t* f;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < kPre) f[tid] = (tid >= 1 && S[tid]) ? 1u : 0u;
    }
};
struct Flags32 {
    const uint8_t* S;
    uint32_t n;
    uint32_t* f;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) f[tid] = S[kPre + tid];
    }
};
struct SumLast {  // out[0] = scan[last] + flag[last]: the total of an exclusive scan, so that the host reads one word;
                  // out[1] = flag[1] (is the block's second position an item start: see hist_hint_)
    const uint32_t *scan, *flag;
    uint32_t last;
    uint32_t* out;
    ORZ_HD void operator()(size_t tid) const {
        if (!tid) { out[0] = scan[last] + flag[last]; out[1] = last >= 1 ? flag[1] : 0; }
    }
};
struct CompactPos32 {
    const uint32_t* flag;
    const uint32_t* scan;
    uint32_t n, off;
    uint32_t* out;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n && flag[tid]) out[scan[tid]] = off + (uint32_t)tid;
    }
};
// sort inputs: (bucket_key(x), x) for history item starts and every new position;
//              (hash2(u-1), u) for u in [P-1, len)
// The reference hashes an item's four bytes when it inserts it (src/matcher.rs:115-121); for the
// last three positions of a block those bytes include the sentinel past the block end, which the
// next block's data later replaces in the window.  Their keys are therefore taken before the slide.
struct TailKeys {
    const uint8_t* win;
    uint32_t len;
    uint32_t* tailkey;  // [3] keys of positions len-3, len-2, len-1
    ORZ_HD void operator()(size_t tid) const {
        if (tid < 3) tailkey[tid] = bucket_key(win, len - 3 + (uint32_t)tid);
    }
};
struct BuildKeys {
    const uint8_t* win;
    const uint32_t* hpos;
    const uint32_t* tailkey;
    uint32_t nhist, n;
    uint32_t *keys, *vals, *kkeys, *kvals;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < (size_t)nhist + n) {
            uint32_t x = tid < nhist ? hpos[tid] : kPre + (uint32_t)(tid - nhist);
            keys[tid] = (x < kPre && x + 3 >= kPre) ? tailkey[x + 3 - kPre] : bucket_key(win, x);
            vals[tid] = x;
        }
        if (tid < (size_t)n + 1) {
            uint32_t u = kPre - 1 + (uint32_t)tid;
            kkeys[tid] = hash2(win, u - 1);
            kvals[tid] = u;
        }
    }
};
struct ScatterSlots {  // idx[pos[j]] = j ; run starts
    const uint32_t* keys;
    const uint32_t* pos;
    uint32_t n;
    uint32_t* idx;
    uint32_t* runstart;
    uint32_t* runend;  // optional
    uint32_t idx_from = 0;  // positions below this one get no idx entry (fast mode: nothing asks for the slot of a history position,
                            // and a scattered 4-byte store costs a 64-byte line: a quarter of the slots are history on text)
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        if (pos[tid] >= idx_from) idx[pos[tid]] = (uint32_t)tid;
        if (tid == 0 || keys[tid] != keys[tid - 1]) runstart[keys[tid]] = (uint32_t)tid;
        if (runend && (tid + 1 == n || keys[tid + 1] != keys[tid])) runend[keys[tid]] = (uint32_t)tid + 1;
    }
};
struct SlotInit {  // history slots carry their final item state, new slots start empty
    const uint8_t* win;
    const uint32_t* epos;
    uint32_t n;
    const uint8_t* ML;
    const uint32_t* ORD;
    SlotRec* srec;
    uint64_t* vbits;  // zeroed
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= n) return;
        uint32_t x = epos[tid];
        if (x < kPre) {
            srec[tid] = SlotRec{x, ORD[x], ML[x], 0, ldu64(win + x), ldu64(win + x + 8)};
            atom_or64(&vbits[tid >> 6], 1ull << (tid & 63));
        } else {
            srec[tid] = SlotRec{x, 0, 255, 0, ldu64(win + x), ldu64(win + x + 8)};
        }
    }
};
struct FillExit {
    uint64_t* exitst;
    uint32_t nseg, seg;
    ORZ_HD void operator()(size_t tid) const {
        if (tid <= nseg) {  // "nothing crosses the segment boundary", sweep stamp 0
            const uint32_t v = ((kPre + (uint32_t)tid * seg) << 2) | kTyLit;
            exitst[tid] = ExitPair::make(0, false, v, v);
        }
    }
};
struct ParseCtlInit {
    ParseCtl* ctl;
    ORZ_HD void operator()(size_t tid) const {
        if (tid) return;
        ctl->front[0] = 0; ctl->front[1] = 0;
        ctl->fchg[0] = kNoChange; ctl->fchg[1] = kNoChange;
        ctl->evals = 0;
        ctl->skipped = 0;
        ctl->nprof = 0;
        ctl->slow = 0;
        ctl->wend = 0;
        for (int i = 0; i < 8; i++) { ctl->prof[i] = 0; ctl->prof2[i] = 0; ctl->prof3[i] = 0; }
        for (int i = 0; i < 16; i++) { ctl->adv_hist[i] = 0; ctl->p1_hist[i] = 0; }
        for (int i = 0; i < 8; i++) ctl->cause[i] = 0;
        for (int i = 0; i < 32; i++) ctl->stop_cause[i] = 0;
    }
};
struct FinalizeBlock {  // slot state -> per-position arrays of the new region
    const uint32_t* idx;
    const uint32_t* kidx;
    const SlotRec* srec;
    const uint64_t* kbits;
    uint32_t len;
    uint8_t *S, *ML, *E;
    uint32_t* ORD;
    ORZ_HD void operator()(size_t tid) const {
        uint32_t x = kPre + (uint32_t)tid;
        if (x >= len) return;
        uint32_t j = idx[x];
        uint32_t m = srec[j].ml;
        S[x] = m != 255;
        ML[x] = m == 255 ? 0 : (uint8_t)m;
        ORD[x] = srec[j].ord;
        uint32_t e = 0;
        if (x >= kPre + 1) {
            uint32_t ks = kidx[x - 2];
            e = (uint32_t)((kbits[ks >> 6] >> (ks & 63)) & 1);
        }
        E[x] = (uint8_t)e;
    }
};
// words[] carried to the next block: per hash2 key the last position of the block whose update stuck
// (src/lz.rs:203,233) = the last set bit of the key's run in the word-predictor bitmap.
struct WordsLastRun {
    const uint64_t *kbits, *k1, *k2;
    const uint32_t *kpos, *krun, *krunend;
    uint32_t* wlast;  // [32768] out: position u, 0 = no update in this block
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= 32768) return;
        const uint32_t lo = krun[tid], hi = krunend[tid];
        uint32_t slot = 0, found = 0, nwords = 0;
        if (hi > lo) found = collect_slots(kbits, k1, k2, kbits[(hi - 1) >> 6], k1[(hi - 1) >> 12], hi, lo, 1, &slot, 1, nwords, [] { return false; });
        wlast[tid] = found ? kpos[slot] : 0;
    }
};
struct ChunkTotals {  // total payload bits of each chunk = header + items
    const uint32_t* bscan;
    const uint32_t* blen;
    const uint32_t* hdrbits;
    uint32_t nitems, nchunks;
    uint32_t* tot;
    ORZ_HD void operator()(size_t tid) const {
        if (tid >= nchunks) return;
        uint32_t i0 = (uint32_t)tid << 20;
        uint32_t i1 = i0 + kChunkItems < nitems ? i0 + kChunkItems : nitems;
        tot[tid] = hdrbits[tid] + (bscan[i1 - 1] + blen[i1 - 1] - bscan[i0]);
    }
};

// ---- the finished stream in DEVICE memory (round 6) -------------------------------------------------------------------------
// Until round 5 the host framed every block: it read the chunk sizes and the guards' flags (one wait), then every chunk's bytes
// (another wait) -- 2 of the 3.83 host waits per block -- and the finished stream existed in host memory only (the multi-GPU
// gather copied it back to the device to send it).  Now the device frames: FrameChunks appends { LEB128(t) chunk[t] }*
// (src/lib.rs:79-80, src/ioutil.rs:79-88) of a block to the stream at the offset the control block holds, behind that block's
// tail stage and gate on the copy stream, and REFUSES to when the block's guards have a finding (the validity gate, the
// ranking guard's second run, a full buffer): the failure is sticky, nothing of that block or a later one is framed, and the
// host learns it when it reads the control block at the end of the stream -- "nothing of this block is handed out" holds.
enum : uint32_t { kOutOk = 0, kOutGate = 1, kOutRank = 2, kOutFull = 3, kOutChunk = 4 };
struct OutCtl {
    unsigned long long off;   // bytes of the stream framed so far
    unsigned long long cap;   // bytes the buffer holds
    uint32_t fail;            // what stopped the framing (kOut*), sticky
    uint32_t fail_block;
    uint32_t redo;            // blocks whose symbol ranking the guard repeated
    uint32_t rankdiff;        // (ORZ_SYMRANK_VERIFY) ranks that differed between two runs
    uint32_t rank_bad[2];     // impossible ranks after the first / second run of the failing block
    uint32_t blocks;          // blocks framed
    uint32_t pad;
    uint32_t gate[kVeCount];  // the gate's findings of the failing block
};
struct FrameLayout {  // what a block appends, from its chunk totals: every thread of FrameChunks and FrameAdvance derives the same
    const uint32_t* tot;      // [nchunks] payload bits of each chunk
    const uint32_t* flags;    // [4 + kVeCount] ranking guard | gate findings (TailSet::srflags)
    uint32_t nchunks;
    uint32_t capwords;        // words a chunk's staging area holds
    ORZ_HD uint32_t verdict(const OutCtl* ctl, uint64_t* total) const {
        if (ctl->fail) return ctl->fail;
        if (flags[1]) return kOutRank;
        for (uint32_t c = 0; c < kVeFirst; c++)
            if (flags[4 + c]) return kOutGate;
        uint64_t t = 0;
        for (uint32_t i = 0; i < nchunks; i++) {
            const uint64_t tb = ((uint64_t)tot[i] + 31) / 32 * 4;  // finish pads to 32 bits, src/coder.rs:75-82
            if (tb / 4 > capwords) return kOutChunk;
            uint64_t v = tb, lenb = 1;
            while (v >= 128) { lenb++; v /= 128; }
            t += lenb + tb;
        }
        *total = t;
        return ctl->off + t > ctl->cap ? kOutFull : kOutOk;
    }
};
struct FrameChunks {  // grid-stride, thread per 16 payload bytes; thread 0 writes the length prefixes
    FrameLayout lay;
    const uint32_t* words;    // [nchunks][capwords] the chunks' payloads (TailSet::out)
    const OutCtl* ctl;
    uint8_t* dst;
    uint32_t nthreads;
    ORZ_HD void operator()(size_t tid) const {
        uint64_t total = 0;
        if (lay.verdict(ctl, &total) != kOutOk) return;
        uint64_t at = ctl->off;
        for (uint32_t i = 0; i < lay.nchunks; i++) {
            const uint64_t tb = ((uint64_t)lay.tot[i] + 31) / 32 * 4;
            if (tid == 0) {  // write_len, src/ioutil.rs:79-88
                uint64_t v = tb, q = at;
                while (v >= 128) { dst[q++] = (uint8_t)(128 + v % 128); v /= 128; }
                dst[q] = (uint8_t)v;
            }
            { uint64_t v = tb; at++; while (v >= 128) { at++; v /= 128; } }
            const uint8_t* src = reinterpret_cast<const uint8_t*>(words + (size_t)i * lay.capwords);
            for (uint64_t u = (uint64_t)tid * 16; u < tb; u += (uint64_t)nthreads * 16) {
                if (u + 16 <= tb) {
                    const uint64_t a = *reinterpret_cast<const uint64_t*>(src + u), b = *reinterpret_cast<const uint64_t*>(src + u + 8);
                    stu64(dst + at + u, a); stu64(dst + at + u + 8, b);
                } else {
                    for (uint64_t k = u; k < tb; k++) dst[at + k] = src[k];
                }
            }
            at += tb;
        }
    }
};
struct FrameAdvance {  // one thread, the launch behind FrameChunks: moves the offset or records why the block was not framed
    FrameLayout lay;
    OutCtl* ctl;
    uint32_t block;
    ORZ_HD void operator()(size_t tid) const {
        if (tid) return;
        if (lay.flags[0]) ctl->redo++;
        ctl->rankdiff += lay.flags[2];
        uint64_t total = 0;
        const uint32_t v = lay.verdict(ctl, &total);
        if (v == kOutOk) { ctl->off += total; ctl->blocks++; return; }
        if (ctl->fail) return;  // (an earlier block's failure stands)
        ctl->fail = v; ctl->fail_block = block;
        ctl->rank_bad[0] = lay.flags[0]; ctl->rank_bad[1] = lay.flags[1];
        for (uint32_t c = 0; c < kVeCount; c++) ctl->gate[c] = lay.flags[4 + c];
    }
};
struct FrameReset {  // a new stream: an empty buffer of `cap` bytes
    OutCtl* ctl;
    unsigned long long cap;
    ORZ_HD void operator()(size_t tid) const {
        if (tid) return;
        ctl->off = 0; ctl->cap = cap; ctl->fail = kOutOk; ctl->fail_block = 0; ctl->redo = 0; ctl->rankdiff = 0;
        ctl->rank_bad[0] = ctl->rank_bad[1] = 0; ctl->blocks = 0; ctl->pad = 0;
        for (uint32_t c = 0; c < kVeCount; c++) ctl->gate[c] = 0;
    }
};
struct FrameEof {  // the EOF chunk: write_len(0), src/lib.rs:89
    OutCtl* ctl;
    uint8_t* dst;
    ORZ_HD void operator()(size_t tid) const {
        if (tid || ctl->fail) return;
        if (ctl->off + 1 > ctl->cap) { ctl->fail = kOutFull; return; }
        dst[ctl->off] = 0;
        ctl->off += 1;
    }
};
struct FrameInject {  // (tests of ORZ_VERIFY=decode) one bit of the stream's first chunk flipped behind every guard
    uint8_t* dst;
    unsigned long long at;
    ORZ_HD void operator()(size_t tid) const {
        if (tid) return;
        uint64_t tb = 0, lenb = 0, sh = 0;
        for (;;) { const uint8_t b = dst[lenb++]; tb |= (uint64_t)(b & 0x7f) << sh; sh += 7; if (!(b & 0x80)) break; }
        if (at < tb) dst[lenb + at] ^= 1;
    }
};
// bytes that always hold the stream of n input bytes: an item costs at most 15 bits (a literal) / 40 bits for at least four
// bytes (a match) / 15 bits for two (WORD), a chunk's three tables and the census header a few kilobytes, LEB128 and padding a few bytes
inline size_t stream_bound(size_t n) { return 2 * n + n / 64 + (1u << 20); }

// Output bytes of an encode call: grows like a vector but does not zero what it is about to receive from the device, and
// hands its malloc'ed buffer to the caller (the C ABI returns it; orz_free releases it) instead of copying it.
class ByteBuf {
   public:
    ByteBuf() = default;
    ~ByteBuf() { std::free(p_); }
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    void reserve(size_t c) {
        if (c <= cap_) return;
        void* q = std::realloc(p_, c);
        if (!q) throw std::bad_alloc();
        p_ = (uint8_t*)q;
        cap_ = c;
    }
    void resize(size_t m) {
        if (m > cap_) reserve(std::max(m, cap_ * 2));
        n_ = m;
    }
    void push_back(uint8_t b) {
        if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 4096);
        p_[n_++] = b;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    void clear() { n_ = 0; }
    uint8_t* release() {  // (never null: an empty result still owns one byte)
        if (!p_) reserve(1);
        uint8_t* q = p_;
        p_ = nullptr;
        n_ = cap_ = 0;
        return q;
    }

   private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

template <class BE>
class StreamEncoder {
   public:
    static constexpr size_t kChunkCapWords = (size_t)kChunkItems * 40 / 32 + 8192;  // payload words per chunk
    static constexpr uint32_t kMaxChunks = 17;
    static constexpr uint32_t kNumKeys = 256 * kHash;
    static constexpr uint32_t kDirtyWords = kNumKeys / 64 + 1;  // one bit per (ctx, hash) run
    static constexpr int kFirstPasses = 12;                     // repair passes queued before the first read-back (lists form; text needs 6..7 under the
                                                                // default schedule, 9..11 under the settled one: round 6 -- a pass behind the one that found nothing is a dozen empty launches)
    static constexpr uint32_t kRepairGrid = 16384;              // threads of the kernels that run over the repair stage's short lists

    // `fast`: the GPU-native parse mode (orz_fast.h) instead of the reference-identical one; `fast_tile` positions
    // per Gauss-Seidel tile (a multiple of 4096), `fast_rounds` rounds per tile
    StreamEncoder(BE& be, Cfg cfg, uint32_t seg_size = 62, uint32_t win_segs = 3072, bool fast = false,
                  uint32_t fast_tile = kFastTile, uint32_t fast_rounds = kFastRounds)
        : be_(be), cfg_(cfg), seg_(seg_size), wsegs_(win_segs), fast_(fast), ftile_(fast_tile), frounds_(fast_rounds) {
        if (fast_) {
            if (ftile_ < kSub || ftile_ % kSub || ftile_ > kNewMax) throw std::runtime_error("fast tile must be a multiple of 4096 in [4096, 16777216]");
            if (frounds_ < 1 || frounds_ > 64) throw std::runtime_error("fast rounds must be in [1, 64]");
            // run predecessors tabulated per position: item starts are about a quarter of a run's positions on text and far
            // fewer in runs of "interior" 4-grams, so the table reaches well beyond 4 x depth
            fK_ = kFastK;  // (deeper runs: the compact lists of final item starts, FastEval / FastRetire)
            sched_auto_ = ftile_ == kFastTile && frounds_ == kFastRounds;
            if (cfg.depth > 20) sched_tile_ = kSettledTileDeep;
            if (const char* sc = getenv("ORZ_FAST_SCHED")) {
                unsigned t = 0, r = 0;
                if (sscanf(sc, "%ux%u", &t, &r) == 2 && t >= kSub && t % kSub == 0 && t <= kNewMax && r >= 1 && r <= 64) { sched_tile_ = t; sched_rounds_ = r; }
                else if (!strcmp(sc, "0")) sched_auto_ = false;
                else throw std::runtime_error("ORZ_FAST_SCHED must be 0 or <tile>x<rounds>");
            }
            if (const char* u = getenv("ORZ_FAST_UNIT")) unit_ = (uint32_t)atoi(u);
            if (unit_ < (1u << 20) || unit_ > kNewMax || unit_ % kSub) throw std::runtime_error("ORZ_FAST_UNIT must be a multiple of 4096 in [1 MiB, 16 MiB]");
            cur_unit_ = unit_;
            if (const char* u = getenv("ORZ_FAST_LEADUNIT")) lead_unit_ = (uint32_t)atoi(u);
            if (lead_unit_ && (lead_unit_ < (1u << 20) || lead_unit_ > kNewMax || lead_unit_ % kSub)) throw std::runtime_error("ORZ_FAST_LEADUNIT must be 0 or a multiple of 4096 in [1 MiB, 16 MiB]");
        }
        if (const char* inj = getenv("ORZ_VERIFY_INJECT")) {  // (tests of the validity gate) "<class>:<n>"
            static const char* names[] = {"", "hole", "context", "ring", "lenmin", "word", "bytes", "lenmin2"};
            const std::string v(inj);
            const size_t colon = v.find(':');
            const std::string cls = v.substr(0, colon);
            for (uint32_t k = 1; k < 8; k++)
                if (cls == names[k]) inject_kind_ = k;
            if (!inject_kind_) throw std::runtime_error("ORZ_VERIFY_INJECT: unknown class");
            inject_nth_ = colon == std::string::npos ? 0 : (uint32_t)strtoul(v.c_str() + colon + 1, nullptr, 10);
        }
        if (const char* oi = getenv("ORZ_OUTPUT_INJECT")) out_inject_ = (size_t)strtoull(oi, nullptr, 10);  // (tests) a flipped bit behind the gate
        // The three fault-injection hooks of the tests live in the shipped path (the GPU tier drives the product library): an
        // encoder that finds one of them set says so on stderr every time it is built -- a stray variable must not pass unnoticed.
        for (const char* hook : {"ORZ_VERIFY_INJECT", "ORZ_OUTPUT_INJECT", "ORZ_SYMRANK_INJECT"})
            if (getenv(hook))
                fprintf(stderr, "orz: WARNING: the test hook %s is set in the environment: this encoder DELIBERATELY DAMAGES what it encodes "
                                "(its streams are invalid unless a guard catches the damage); unset it for real use\n", hook);
        if (seg_ < 8 || seg_ > kSegMax) throw std::runtime_error("seg_size must be in [8, 62]");
        if (wsegs_ < 1) throw std::runtime_error("window must hold at least one segment");
        try {
            dmax_ = (uint32_t)std::max(cfg.depth, std::max(cfg.lazy1, cfg.lazy2));
            if (cfg.depth < 1 || dmax_ > 200) throw std::runtime_error("LZCfg depth out of range");
            nseg_max_ = (kNewMax + seg_ - 1) / seg_;
            if (wsegs_ > nseg_max_) wsegs_ = nseg_max_;
            ring_ = wsegs_ + 8;
            winbuf_ = take<uint8_t>((size_t)kBlock + 2 * kSent + 64);
            S_ = take<uint8_t>(kWLen);
            if (!fast_) E_ = take<uint8_t>(kWLen);  // (exact mode only, like LR_)
            ML_ = take<uint8_t>(kWLen);
            ORD_ = take<uint32_t>(kWLen);
            if (!fast_) LR_ = take<uint8_t>(kWLen);
            SRC_ = take<uint32_t>(kWLen, false);
            W0_ = take<uint8_t>(kWLen);
            TY_ = take<uint8_t>(kWLen);
            LENMIN_ = take<uint8_t>(kWLen);
            LMV_ = take<uint8_t>(kWLen);
            idx_ = take<uint32_t>(kWLen, false);
            kidx_ = take<uint32_t>(kWLen, false);
            // sort buffers double as u64 scratch of the post stage (entA_/entB_ views)
            entA_ = take<uint64_t>((size_t)kWLen, false);
            entB_ = take<uint64_t>((size_t)kWLen, false);
            epos_ = take<uint32_t>(kWLen, false);
            kpos_ = take<uint32_t>((size_t)kNewMax + 8);
            runstart_ = take<uint32_t>(kNumKeys + 1);
            krun_ = take<uint32_t>(32768 + 1);
            krunend_ = take<uint32_t>(32768 + 1);
            vbits_ = take<uint64_t>(kWLen / 64 + 2);
            v1_ = take<uint64_t>(kWLen / 4096 + 2);
            v2_ = take<uint64_t>(kWLen / 262144 + 2);
            kbits_ = take<uint64_t>(kNewMax / 64 + 2);
            k1_ = take<uint64_t>(kNewMax / 4096 + 2);
            k2_ = take<uint64_t>(kNewMax / 262144 + 2);
            if (!fast_) {
                srec_ = take<SlotRec>(kWLen);
                exitst_ = take<uint64_t>((size_t)nseg_max_ + 2);
                hist_ = take<uint8_t>((size_t)ring_ * 256);
                base_ = take<uint32_t>((size_t)ring_ * 256);
                ctl_ = take<ParseCtl>(1);
                partial_ = take<uint32_t>((size_t)2 * (wsegs_ / kRankChunk + 1) * 256);
            } else {
                const size_t nn = (size_t)kNewMax + 512;
                frows_ = take<uint8_t>((size_t)kNewMax * fK_ + 64, false);
                frlen_ = take<uint8_t>(nn);
                fstext_ = take<uint64_t>((size_t)kWLen * 2, false);
                fkw_ = take<uint16_t>(nn);
                fev_ = take<uint32_t>(nn, false);
                fty_ = take<uint8_t>(nn);
                fnl_ = take<uint8_t>(nn);
                fpt_ = take<uint8_t>(nn);
                fmf_ = take<uint8_t>(nn);
                fef_ = take<uint8_t>(nn);
                fdirty_ = take<uint8_t>(nn);
                frdist_ = take<uint64_t>(nn, false);
                fwmask_ = take<uint64_t>(nn, false);
                fkmeta_ = take<uint16_t>(nn);
                fhz_ = take<uint32_t>((size_t)(kNSub + 2) * 256 * 4);
                fhcm_ = take<uint32_t>((size_t)kHistSub * 256);
                fhpre_ = take<uint32_t>((size_t)(kHistSub + 1) * 256);
                fgsum_ = take<uint32_t>((size_t)(kNSub / 64 + 4) * 256);
                fdiag_ = take<unsigned long long>(64);  // diagnostics counters (a.stats) and FastVerify's findings: an allocation of their own
                fx0_ = take<uint8_t>(nn);
                fx1_ = take<uint8_t>((size_t)(kNSub + 2) * kEntries);
                fx2_ = take<uint8_t>((size_t)(kNewMax / kSub + 4) * kEntries);  // (sized for the finest tile)
                fsbits_ = take<uint64_t>(kNewMax / 64 + 16);
                fcentry_ = take<uint32_t>(kNSub + 2);
                ftentry_ = take<uint32_t>(kNewMax / kSub + 4);
                fcm_ = take<uint32_t>((size_t)(kNSub + 2) * 256);
                fcp_ = take<uint32_t>((size_t)(kNSub + 2) * 256);
                fcut_ = take<uint32_t>(nn);
                frdirty_ = take<uint64_t>((size_t)2 * kDirtyWords);
                flaste_ = take<uint32_t>(nn);
                fctl_ = take<FastCtl>(1);
                fcok_ = take<uint32_t>(512);
                ffarv_ = take<uint32_t>(nn, false);
                fcl_ = take<uint64_t>((size_t)kWLen * 2, false);
                fccnt_ = take<uint32_t>(kNumKeys + 1);
                fcnew_ = take<uint32_t>(kNumKeys + 1);
            }
            f32_ = take<uint32_t>((size_t)kNewMax + 8, false);   // flags / scan over at most 2^24 + 1 entries (the history's
            sc32_ = take<uint32_t>((size_t)kNewMax + 8, false);  // positions, the new positions, the word-list slots)
            hpos_ = take<uint32_t>(kPre + 1);
            ctxcount_ = take<uint32_t>(256);
            tailkey_ = take<uint32_t>(8);
            wsnap_ = take<uint8_t>(65536);
            wlast_ = take<uint32_t>(32768);
            // items and tail-stage buffers: two sets, taken by the blocks alternately (see post_stage)
            for (TailSet& t : ts_) {
                t.cap = 0;
                // (ORZ_TAIL_ITEMS0: tests of the growth path start small)
                grow_tail_set(t, getenv("ORZ_TAIL_ITEMS0") ? std::max<uint32_t>(65536u, (uint32_t)strtoul(getenv("ORZ_TAIL_ITEMS0"), nullptr, 10)) : kTailItems0);
                t.rstart = take<uint32_t>(520);
                t.hw = take<uint32_t>((size_t)kMaxChunks * kHwStride);
                t.hl = take<uint8_t>((size_t)kMaxChunks * kHwStride);
                t.hc = take<uint16_t>((size_t)kMaxChunks * kHwStride);
                t.hdrbits = take<uint32_t>(kMaxChunks);
                t.tot = take<uint32_t>(kMaxChunks);
                t.srflags = take<uint32_t>(4 + kVeCount);  // (+ the gate's findings of the block: read with the flags in one copy)
                t.out = take<uint32_t>((size_t)kMaxChunks * kChunkCapWords, false);
            }
            // the validity gate's own decoder state (orz_verify.h)
            vrec_ = take<uint32_t>(kWLen);
            vord_ = take<uint32_t>(kWLen, false);
            vctx_ = take<uint32_t>(256);
            vlast_ = take<uint32_t>(2);
            vwords_ = take<uint8_t>(65536);
            counts_ = take<uint32_t>(kSyms + 3);
            order_ = take<uint16_t>(kSyms + 3);
            ncounted_ = take<uint32_t>(4);
            srstate_ = take<uint16_t>((size_t)512 * kSrWords);
            srbackup_ = take<uint16_t>((size_t)512 * kSrWords);
            outoff_ = take<uint64_t>(kMaxChunks);
            octl_ = take<OutCtl>(1);
            {
                std::vector<uint64_t> off(kMaxChunks);
                for (uint32_t i = 0; i < kMaxChunks; i++) off[i] = (uint64_t)i * kChunkCapWords;
                be_.h2d(outoff_, off.data(), kMaxChunks * 8);
            }
            reset();
        } catch (...) {  // a failed allocation / launch must not leak the earlier ones
            release_all();
            throw;
        }
    }
    ~StreamEncoder() { release_all(); }
    StreamEncoder(const StreamEncoder&) = delete;
    StreamEncoder& operator=(const StreamEncoder&) = delete;

    // LZEncoder::new (src/lz.rs:75-80): empty rings, zero word table, after_literal = true
    void reset() {
        be_.memset(winbuf_, 0, (size_t)kBlock + 2 * kSent + 64);
        be_.memset(S_, 0, kWLen);
        be_.memset(ML_, 0, kWLen);
        be_.memset(ORD_, 0, (size_t)kWLen * 4);
        be_.memset(LENMIN_, 0, kWLen);
        be_.memset(ctxcount_, 0, 256 * 4);
        be_.memset(wsnap_, 0, 65536);
        be_.memset(vrec_, 0, (size_t)kPre * 4);  // (no item starts in the history of a new stream)
        be_.memset(vwords_, 0, 65536);
        be_.launch(256, VerReset{vctx_, vlast_});
        // whatever an earlier stream left on the side streams (an encode that failed half-way; a finished one left nothing) is
        // ordered BEFORE this stream's work without the host waiting for it (round 6: these were three host waits per stream):
        // the main stream waits for the side streams, and the side streams start every block behind an event of the main one
        for (int k = 1; k <= 3; k++) { be_.select(k); be_.record(kEvIdle + k - 1); }
        be_.select(0);
        for (int k = 1; k <= 3; k++) be_.wait(kEvIdle + k - 1);
        for (TailSet& t : ts_) { t.pending = false; t.framed = false; }
        dev_out_ = false;
        pend_order_.clear();
        cur_set_ = 0;
        if (fast_) {
            // Round state that no parse resets because every entry is written before it is read: the emulation fills it
            // with garbage here (HIP: nothing), so that a read of something left over by an earlier stream cannot hide --
            // a reused encoder must write what a fresh one writes (tests/test_emu_fast.py)
            const size_t nn = (size_t)kNewMax + 512;
            be_.poison(fhz_, (size_t)(kNSub + 2) * 256 * 4 * 4);
            be_.poison(fdirty_, nn);
            be_.poison(ffarv_, nn * 4);
            be_.poison(fev_, nn * 4);
        }
        lt_carry_ = kTyLit;
        settled_next_ = false;
        hist_hint_ = ~0u;
        if (fast_) { const uint32_t lt = kTyLit; be_.h2d(&fctl_->lt, &lt, 4); }
        stream_start_ = true;
        stats = EncodeStats();
    }

    uint8_t* dwin() { return winbuf_ + kSent; }  // device address of window offset 0
    uint8_t* dwinbuf() { return winbuf_; }       // device address of the allocation (sentinel included)
    uint32_t seg_size() const { return seg_; }
    uint32_t window_segs() const { return wsegs_; }
    bool fast() const { return fast_; }
    uint32_t fast_tile() const { return ftile_; }
    void set_lead_block(bool on) { lead_block_ = on; }  // the next block leads a multi-block stream (see fast_parse)
    uint32_t fast_rounds() const { return frounds_; }
    uint32_t unit_bytes() const { return fast_ ? unit_ : kNewMax; }
    uint32_t fast_row() const { return fK_; }

    // Encode the block whose n new bytes sit at dwin()[kPre, kPre+n).  Appends
    // { LEB128(t) chunk[t] }* (src/lib.rs:76-82, src/ioutil.rs:79-88) to `out`; optionally reports
    // each chunk's end position (the value LZEncoder::encode returns, src/lz.rs:268,346).
    // The last stages of a block (symbol ranking; Huffman + bit pack) run on the backend's streams 1 and 2 and overlap
    // the next blocks' prep + parse; a block's output is appended two calls later, or by finish().
    template <class OutT>
    void encode_block(uint32_t n, OutT& out, std::vector<size_t>* chunk_ends = nullptr) {
        if (n == 0 || n > kNewMax) throw std::runtime_error("bad block size");
        struct TokenGuard {  // (members jobs: at most ORZ_PARSE_TOKENS encoders parse at a time, see the backend: the token goes
            StreamEncoder& e;  // back when the block's items are handed to the ranking chain -- post_stage -- or on the way out)
            explicit TokenGuard(StreamEncoder& x) : e(x) { e.be_.parse_token_acquire(); e.token_held_ = true; }
            ~TokenGuard() { e.release_token(); }
        } token{*this};
        last_n_ = n;
        double t0 = be_.now();
        const uint8_t* win = dwin();
        const uint32_t len = kPre + n;
        const uint32_t nseg = (n + seg_ - 1) / seg_;
        // ---- history item starts -> hpos
        uint32_t nhist = 0;
        if (!stream_start_) {
            be_.launch(kPre, HistFlags32{S_, f32_});
            be_.exclusive_scan_u32(f32_, sc32_, kPre);
            be_.launch(1, SumLast{sc32_, f32_, kPre - 1, tailkey_ + 3});
            be_.launch(kPre, CompactPos32{f32_, sc32_, kPre, 0, hpos_});
            // After a slide by a whole block the history IS the block before: its items but the first (it left the window)
            // and one at the second position (it sits at offset 0 now, which is dead, src/matcher.rs:85) -- the host knows
            // that count since it read the block's item total: no read-back, no synchronisation here.
            if (hist_hint_ != ~0u && !be_.check_hints()) nhist = hist_hint_;
            else {
                be_.d2h(&nhist, tailkey_ + 3, 4);
                if (hist_hint_ != ~0u && hist_hint_ != nhist) throw std::runtime_error("history item count differs from the host's arithmetic");
            }
            hist_hint_ = ~0u;
        }
        // ---- candidate lists: stable radix sort of positions by (ctx8, hash) and by hash2
        const uint32_t nent = nhist + n;
        uint32_t* keysA = (uint32_t*)entA_;
        uint32_t* valsA = keysA + kWLen;
        uint32_t* keysB = (uint32_t*)entB_;
        uint32_t* kkeysA = f32_;
        uint32_t* kvalsA = sc32_;
        uint32_t* kkeysB = keysB + kWLen;
        be_.launch(std::max<size_t>(nent, (size_t)n + 1), BuildKeys{win, hpos_, tailkey_, nhist, n, keysA, valsA, kkeysA, kvalsA});
        be_.sort_pairs_u32(keysA, keysB, valsA, epos_, nent, 21);
        be_.launch(nent, ScatterSlots{keysB, epos_, nent, idx_, runstart_, nullptr, fast_ ? kPre : 0u});
        be_.sort_pairs_u32(kkeysA, kkeysB, kvalsA, kpos_, (size_t)n + 1, 15);
        if (fast_) {
            // every reset the prep and the post stage need, ONE launch (round 6; before: eight fill dispatches)
            ZeroRanges z;
            z.add(krun_, 32769 * 4);
            z.add(krunend_, 32769 * 4);
            z.add(vbits_ + nent / 64, 16);  // (the words behind the last slot; FastSlotInitWave writes the others whole)
            z.add(LENMIN_ + kPre, kWLen - kPre);
            z.add(fhpre_, 256 * 4);
            z.add(fccnt_, (size_t)(kNumKeys + 1) * 4);
            if (stream_start_) z.add(fhcm_, (size_t)kHistSub * 256 * 4);
            be_.launch(z.units(), z);
        } else {
            be_.memset(krun_, 0, 32769 * 4);
            be_.memset(krunend_, 0, 32769 * 4);
        }
        be_.launch((size_t)n + 1, ScatterSlots{kkeysB, kpos_, n + 1, kidx_, krun_, krunend_});
        if (fast_) {
            double tf = be_.now();  // (no sync here: the prep kernels are queued, the parse follows on the same stream)
            stats.t_prep += tf - t0;
            fast_parse(n, len, nent, keysB, kkeysB, out);
            double t2f = be_.now();
            stats.t_parse += t2f - tf;
            post_stage(n, len, out, chunk_ends, t2f);
            return;
        }
        be_.memset(vbits_, 0, ((size_t)nent / 64 + 1) * 8);
        be_.memset(v1_, 0, ((size_t)kWLen / 4096 + 2) * 8);
        be_.memset(v2_, 0, ((size_t)kWLen / 262144 + 2) * 8);
        be_.memset(kbits_, 0, ((size_t)n / 64 + 2) * 8);
        be_.memset(k1_, 0, ((size_t)kNewMax / 4096 + 2) * 8);
        be_.memset(k2_, 0, ((size_t)kNewMax / 262144 + 2) * 8);
        be_.launch(nent, SlotInit{win, epos_, nent, ML_, ORD_, srec_, vbits_});
        // summary levels of both bitmaps, exact before the first sweep (0 rank chunks = rebuild only)
        be_.rank(RankArgs{win, ctl_, hist_, base_, partial_, idx_, srec_, LR_, nseg, seg_, wsegs_, ring_, len, 0, vbits_, kbits_, v1_, v2_, k1_, k2_, nent / 64 + 1, n / 64 + 2, nullptr}, 0);
        be_.launch((size_t)nseg + 1, FillExit{exitst_, nseg, seg_});
        be_.memset(hist_, 0, (size_t)ring_ * 256);
        be_.d2d(base_, ctxcount_, 256 * 4);
        be_.launch(1, ParseCtlInit{ctl_});
        be_.memset(partial_, 0, (size_t)2 * (wsegs_ / kRankChunk + 1) * 256 * 4);
        be_.memset(LENMIN_ + kPre, 0, kWLen - kPre);
        be_.sync();
        double t1 = be_.now();
        stats.t_prep += t1 - t0;

        // ---- speculative sweeps to the causal fixed point
        ParseArgs pa;
        pa.win = win; pa.len = len; pa.nseg = nseg; pa.seg = seg_; pa.wsegs = wsegs_; pa.ring = ring_;
        pa.depth = (uint32_t)cfg_.depth; pa.lazy1 = (uint32_t)cfg_.lazy1; pa.lazy2 = (uint32_t)cfg_.lazy2; pa.dmax = dmax_;
        pa.lt0 = lt_carry_; pa.par = 0; pa.prof = (getenv("ORZ_PROF") ? 1 : 0) | (getenv("ORZ_NO_E1") ? 2 : 0);
        pa.chain = getenv("ORZ_CHAIN") ? (uint32_t)atoi(getenv("ORZ_CHAIN")) : 24;
        if (pa.chain < 1) pa.chain = 1;
        pa.polls = getenv("ORZ_POLLS") ? (uint32_t)atoi(getenv("ORZ_POLLS")) : be_.handoff_polls();
        if (getenv("ORZ_MAXPASS")) pa.maxpass = (uint32_t)atoi(getenv("ORZ_MAXPASS"));
        pa.deadline = getenv("ORZ_DEADLINE_US") ? (uint32_t)atoi(getenv("ORZ_DEADLINE_US")) * 100 : be_.handoff_deadline();
        pa.near = getenv("ORZ_NEAR") ? (uint32_t)atoi(getenv("ORZ_NEAR")) : be_.near_blocks();
        pa.far_deadline = getenv("ORZ_FAR_DEADLINE_US") ? (uint32_t)atoi(getenv("ORZ_FAR_DEADLINE_US")) * 100 : be_.far_deadline();
        pa.skip_after = getenv("ORZ_SKIP_US") ? (uint32_t)atoi(getenv("ORZ_SKIP_US")) * 100 : be_.skip_after();
        pa.skip_rand = getenv("ORZ_SKIP_RAND") ? (uint32_t)atoi(getenv("ORZ_SKIP_RAND")) : 0;
        pa.srec = srec_; pa.idx = idx_; pa.runstart = runstart_; pa.kpos = kpos_; pa.kidx = kidx_; pa.krun = krun_;
        pa.wsnap = wsnap_; pa.vbits = vbits_; pa.v1 = v1_; pa.v2 = v2_; pa.kbits = kbits_; pa.k1 = k1_; pa.k2 = k2_; pa.exitst = exitst_;
        pa.hist = hist_; pa.base = base_; pa.TY = TY_; pa.SRC = SRC_; pa.W0 = W0_; pa.LR = LR_; pa.partial = partial_; pa.ctl = ctl_;
        pa.sig = nullptr;
        if (getenv("ORZ_PROF")) {  // diagnostics only: borrow the (idle during the parse) scan buffer
            pa.sig = sc32_;
            be_.memset(sc32_, 0, (size_t)(nseg + 1) * 16);
        }
        unsigned long long* tim = nullptr;
        if (getenv("ORZ_TIMELINE")) {  // diagnostics only: wall-clock stamps of every wave of one sweep -> stderr
            tim = be_.template alloc<unsigned long long>((size_t)wsegs_ * 8);
            be_.memset(tim, 0, (size_t)wsegs_ * 64);
            pa.tim = tim; pa.timsweep = (uint32_t)atoi(getenv("ORZ_TIMELINE"));
        }
        const size_t lds_bytes = ParseLds::make(dmax_, pa.prof & 1).total;
        const uint32_t grid = std::min(wsegs_, nseg);
        uint32_t par = 0, front = 0, batch = 8;
        uint64_t sweeps = 0;
        while (front < nseg) {
            for (uint32_t i = 0; i < batch; i++) {
                pa.par = par;
                pa.sweep = (uint32_t)sweeps + 1;
                be_.timed_begin();
                be_.launch_waves(grid, ParseWave{pa}, lds_bytes);
                be_.timed_end();
                be_.rank(RankArgs{win, ctl_, hist_, base_, partial_, idx_, srec_, LR_, nseg, seg_, wsegs_, ring_, len, par, vbits_, kbits_, v1_, v2_, k1_, k2_, nent / 64 + 1, n / 64 + 2, pa.sig},
                         wsegs_ / kRankChunk + 1);
                par ^= 1;
                sweeps++;
            }
            ParseCtl h;
            be_.d2h(&h, ctl_, sizeof h);
            const uint32_t adv = h.front[par] > front ? h.front[par] - front : 1;
            front = h.front[par];
            stats.seg_evals = stats.seg_evals + 0;  // (evals are read once, below)
            // size the next batch to what the front still has to cover, assuming this batch's speed
            const uint64_t left = nseg > front ? nseg - front : 0;
            const uint64_t per = std::max<uint64_t>(1, adv / batch);
            batch = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(2, left / per + 1));
            if (getenv("ORZ_TRACE_SWEEPS"))
                fprintf(stderr, "sweeps=%llu front=%u/%u next batch=%u\n", (unsigned long long)sweeps, front, nseg, batch);
        }
        {
            ParseCtl h;
            be_.d2h(&h, ctl_, sizeof h);
            stats.seg_evals += h.evals;
            if (getenv("ORZ_PROF")) {
                fprintf(stderr, "changed segments within 256 of the front: %u ; entry moved %u, words answers moved %u, candidate lists moved %u, none of these %u\n", h.cause[0], h.cause[1], h.cause[2], h.cause[3], h.cause[4]);
                fprintf(stderr, "front stoppers by what had moved (E entry, W words, C candidates): ");
                for (int i = 0; i < 32; i++) if (h.stop_cause[i]) fprintf(stderr, "%s%s%s%s:%u ", (i & 2) ? "E" : "", (i & 4) ? "W" : "", (i & 8) ? "C" : "", (i & 16) ? "first" : ((i & 14) ? "" : "none"), h.stop_cause[i]);
                fprintf(stderr, "\n");
                if (h.prof3[6]) fprintf(stderr, "slow waves (%llu): own-count %llu first-loads %llu slot-walk %llu word-walk %llu records+lcp %llu\n", h.prof3[6], h.prof3[0] / h.prof3[6], h.prof3[1] / h.prof3[6], h.prof3[2] / h.prof3[6], h.prof3[3] / h.prof3[6], h.prof3[4] / h.prof3[6]);
                fprintf(stderr, "phase-1 end, cycles/16K histogram: ");
                for (int i = 0; i < 16; i++) fprintf(stderr, "%u ", h.p1_hist[i]);
                fprintf(stderr, "\nfront advance per sweep (segments): ");
                for (int i = 0; i < 16; i++) fprintf(stderr, "[%d..%d]:%u ", (1 << i) - 1, (2 << i) - 2, h.adv_hist[i]);
                fprintf(stderr, "\n");
            }
            if (getenv("ORZ_PROF") && h.nprof)
                fprintf(stderr, "parse phases (avg shader cycles / sampled wave, %u waves): load %llu  candidates %llu  decide %llu  walk %llu  publish %llu ; slow items %u ; phase-1 slowest-lane stamps: own-count %llu  first-loads %llu  slot-walk %llu  word-walk %llu  records+lcp %llu ; max-lane bitmap words %llu\n",
                        h.nprof, h.prof[0] / h.nprof, h.prof[1] / h.nprof, h.prof[2] / h.nprof, h.prof[3] / h.nprof, h.prof[4] / h.nprof, h.slow, h.prof2[0] / h.nprof, h.prof2[1] / h.nprof, h.prof2[2] / h.nprof, h.prof2[3] / h.nprof, h.prof2[4] / h.nprof, h.prof2[7] / h.nprof);
        }
        if (tim) {
            std::vector<unsigned long long> h((size_t)wsegs_ * 8);
            be_.d2h(h.data(), tim, h.size() * 8);
            unsigned long long t0 = ~0ull;
            for (uint32_t i = 0; i < wsegs_; i++) if (h[i * 8] && h[i * 8] < t0) t0 = h[i * 8];
            if (t0 != ~0ull) {
                fprintf(stderr, "timeline of sweep %u (10 ns ticks from the first wave's start): block seg start p1end walk1 walkend end passes polls changed\n", pa.timsweep);
                for (uint32_t i = 0; i < wsegs_; i++) {
                    const unsigned long long* t = &h[i * 8];
                    if (!t[0]) continue;
                    fprintf(stderr, "TL %u %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", i, t[6], t[0] - t0, t[1] - t0, t[2] ? t[2] - t0 : 0, t[3] - t0, t[4] - t0, t[5] >> 32, t[5] & 0xffffffffu, t[7]);
                }
            }
            be_.free(tim);
        }
        stats.sweeps += sweeps;
        be_.launch(n, FinalizeBlock{idx_, kidx_, srec_, kbits_, len, S_, ML_, E_, ORD_});
        be_.sync();
        double t2 = be_.now();
        stats.t_parse += t2 - t1;
#ifdef ORZ_DEBUG_ROBUST
        {   // consistency of the final ring ordinals: ORD[x] == number of earlier items of the same ctx
            std::vector<uint8_t> hS(len), hw(len + 8);
            std::vector<uint32_t> hO(len), cnt(256, 0), hb(256);
            be_.d2h(hS.data(), S_, len); be_.d2h(hO.data(), ORD_, (size_t)len * 4); be_.d2h(hw.data(), win, len);
            be_.d2h(cnt.data(), ctxcount_, 256 * 4);
            int bad = 0;
            for (uint32_t x = kPre; x < len; x++) if (hS[x]) {
                uint32_t c = (hw[x - 1] & 0x7f) | ((uint32_t)is_alnum(hw[x - 2]) << 7);
                if (hO[x] != cnt[c] && bad++ < 10) fprintf(stderr, "ORD mismatch x=%u seg=%u ctx=%u ord=%u want=%u\n", x, (x - kPre) / seg_, c, hO[x], cnt[c]);
                cnt[c]++;
            }
            fprintf(stderr, "ORD check: %d bad\n", bad);
            if (const char* dp = getenv("ORZ_DUMP_ITEMS")) {
                std::vector<uint8_t> hT(len), hM(len); std::vector<uint32_t> hR(len);
                be_.d2h(hT.data(), TY_, len); be_.d2h(hM.data(), ML_, len); be_.d2h(hR.data(), SRC_, (size_t)len * 4);
                FILE* f = fopen(dp, "w");
                for (uint32_t x = kPre; x < len; x++) if (hS[x]) fprintf(f, "%u ty=%u ml=%u src=%u ord=%u\n", x, hT[x] & 3, hM[x], (hT[x] & 3) == 2 ? hR[x] : 0, hO[x]);
                fclose(f);
            }
        }
#endif

        post_stage(n, len, out, chunk_ends, t2);
    }

    // exclusive prefix down the 256 columns of in[rows][256] -> out[rows + 1][256] (out[0] holds the base)
    // (`ctl`: the repair passes' scans return at once when the passes are done)
    void col_scan(const uint32_t* in, uint32_t rows, uint32_t* out, const FastCtl* ctl = nullptr) {
        const size_t groups = (rows + 63) / 64;
        be_.launch(groups * 256, ColScanGroups{in, rows, fgsum_, ctl});
        be_.launch(256, ColScanTop{fgsum_, rows, out, ctl});
        be_.launch(groups * 256, ColScanRows{in, fgsum_, rows, out, ctl});
    }

    // Diagnostics (ORZ_DEBUG_DUMP=<dir>, ORZ_DEBUG_POS=<window offset>): the per-position state of the fast parse in a window of
    // 512 positions around the offset, written after the rounds ("r") and after the repairs + commit ("f") of every block.
    void debug_dump(const char* tag, uint32_t n, const FastCtl* h) {
        static const char* dir = getenv("ORZ_DEBUG_DUMP");
        if (!dir) return;
        static const uint32_t at = getenv("ORZ_DEBUG_POS") ? (uint32_t)strtoul(getenv("ORZ_DEBUG_POS"), nullptr, 10) : kPre + 256;
        const uint32_t lo = (std::max(at, kPre + 256) - 256) & ~63u, w = 512;
        if (lo + w > kPre + n) return;
        const uint32_t i0 = lo - kPre;  // (a multiple of 64 + 1 ... the bitmap words are fetched around it)
        struct Rec { uint32_t lo, w, block, passes, total, nmem, lastflips, pad; } rec{lo, w, (uint32_t)stats.blocks, h ? h->passes : 0, h ? h->total : 0, h ? h->nmem : 0, h ? h->lastflips : 0, 0};
        std::vector<uint8_t> b8(w);
        std::vector<uint32_t> b32(w);
        std::vector<uint64_t> b64(w / 64 + 2);
        char name[512];
        static std::atomic<unsigned> serial{0};
        snprintf(name, sizeof name, "%s/dump_%05u_%p_b%u_%s.bin", dir, serial.fetch_add(1), (void*)this, (unsigned)stats.blocks, tag);
        FILE* f = fopen(name, "wb");
        if (!f) return;
        fwrite(&rec, sizeof rec, 1, f);
        be_.d2h(b64.data(), fsbits_ + i0 / 64, b64.size() * 8); fwrite(b64.data(), 8, b64.size(), f);
        const uint8_t* a8[] = {fty_ + i0, fnl_ + i0, fpt_ + i0, fmf_ + i0, fef_ + i0, fdirty_ + i0, S_ + lo, TY_ + lo, ML_ + lo, W0_ + lo, LENMIN_ + lo, LMV_ + lo, dwin() + lo};
        for (const uint8_t* q : a8) { be_.d2h(b8.data(), q, w); fwrite(b8.data(), 1, w, f); }
        const uint32_t* a32[] = {SRC_ + lo, ORD_ + lo, fev_ + i0, ffarv_ + i0, idx_ + lo, fcut_ + i0};
        for (const uint32_t* q : a32) { be_.d2h(b32.data(), q, (size_t)w * 4); fwrite(b32.data(), 4, w, f); }
        fclose(f);
    }

    // The GPU-native parse of one block (orz_fast.h): fills S_/TY_/ML_/SRC_/ORD_/W0_ for the new region and
    // carries ctxcount_ / wsnap_ / lt_carry_, like the exact mode's sweeps + FinalizeBlock do.
    template <class OutT>
    void fast_parse(uint32_t n, uint32_t len, uint32_t nent, const uint32_t* slot_keys, const uint32_t* word_keys, OutT& out) {
        const uint8_t* win = dwin();
        const uint32_t nk = n + 1, K = fK_, nsub = (n + kSub - 1) / kSub, nvw = nent / 64 + 2;  // nvw: words of the item-start bitmap
        be_.launch(nk, FastKw{win, kpos_, nk, fkw_});
        be_.launch_waves(((size_t)nk + 63) / 64, FastWordMasks{kpos_, word_keys, krun_, fkw_, wsnap_, nk, fwmask_, fkmeta_}, FastWordMasks::lds_bytes());
        // history item starts per (unified subtile, ctx) and their prefix: what the ring horizons reach back into
        if (!stream_start_) be_.launch_waves(kHistSub, HistCountWave{win, S_, fhcm_}, HistCountWave::lds_bytes());  // (a new stream: zeroed in encode_block)
        col_scan(fhcm_, kHistSub, fhpre_);
        uint64_t* stext = fstext_;
        {
            FastSlotInitWave si{epos_, slot_keys, runstart_, nent, vbits_, frlen_};
            si.win = win; si.stext = stext; si.cl = fcl_; si.ccnt = fccnt_;
            be_.launch_waves(((size_t)nent + 63) / 64, si, 0);
        }
        be_.timed_begin(2);
        be_.launch_waves(((size_t)nent + 63) / 64, FastRowsWave{win, epos_, stext, frlen_, nent, K, frows_, frdist_}, FastRowsWave::lds_bytes(K));
        be_.timed_end(2);
        FastArgs a;
        a.win = win; a.len = len; a.n = n; a.K = K; a.depth = (uint32_t)cfg_.depth; a.lazy1 = (uint32_t)cfg_.lazy1;
        a.lazy2 = (uint32_t)cfg_.lazy2; a.tile = ftile_; a.dmax = dmax_; a.nent = nent; a.nk = nk;
        a.idx = idx_; a.epos = epos_; a.kidx = kidx_; a.kpos = kpos_; a.krun = krun_; a.rows = frows_; a.rlen = frlen_;
        a.rdist = frdist_; a.wmask = fwmask_; a.kmeta = fkmeta_; a.hpre = fhpre_;
        a.kw = fkw_; a.wsnap = wsnap_; a.ORD = ORD_; a.stext = fstext_; a.runstart = runstart_; a.farv = ffarv_;
        a.far = getenv("ORZ_FAST_FAR") ? (uint32_t)atoi(getenv("ORZ_FAST_FAR")) : 16384; a.vbits = vbits_; a.kbits = kbits_; a.v1 = v1_; a.ev = fev_;
        a.ty = fty_; a.nl = fnl_; a.pt = fpt_; a.sbits = fsbits_; a.mfb = fmf_; a.efb = fef_; a.dirty = fdirty_; a.hz = fhz_;
        a.x0 = fx0_; a.x1 = fx1_; a.x2 = fx2_;
        a.cl = fcl_; a.ccnt = fccnt_; a.cnew = fcnew_; a.rounds = frounds_;
        // Repair stage over lists (round 5, orz_fast.h): its lists live in the first sort buffer, which is idle between the prep's
        // sorts and the post stage's.  ORZ_FAST_REPAIR=grid: every pass as grids over all positions (the form of rounds 2-4).
        static const bool repair_lists = !(getenv("ORZ_FAST_REPAIR") && !strcmp(getenv("ORZ_FAST_REPAIR"), "grid"));
        const uint32_t nsub_max = kNewMax / kSub + 2;
        uint8_t* scratch = reinterpret_cast<uint8_t*>(entA_);
        auto carve = [&](size_t bytes) { uint8_t* q = scratch; scratch += (bytes + 255) & ~(size_t)255; return q; };
        uint16_t* mlist = reinterpret_cast<uint16_t*>(carve((size_t)nsub_max * kSubMatches * 2));
        uint16_t* wlist = reinterpret_cast<uint16_t*>(carve((size_t)nsub_max * kSubWords * 2));
        uint32_t* mcnt = reinterpret_cast<uint32_t*>(carve((size_t)nsub_max * 4));
        uint32_t* wcnt = reinterpret_cast<uint32_t*>(carve((size_t)nsub_max * 4));
        uint32_t* cutlist = reinterpret_cast<uint32_t*>(carve(((size_t)kNewMax / kMinLen + 64) * 4));
        uint32_t* fixlist = reinterpret_cast<uint32_t*>(carve(((size_t)kNewMax / 2 + 64) * 4));
        uint32_t* wextra = reinterpret_cast<uint32_t*>(carve(((size_t)kNewMax / 2 + 64) * 4));
        const size_t tbits_bytes = ((size_t)kNewMax / 64 + 8) * 8, kdirty_bytes = 512 * 8;
        uint64_t* tbits = reinterpret_cast<uint64_t*>(carve(tbits_bytes));
        uint64_t* kdirty = reinterpret_cast<uint64_t*>(carve(kdirty_bytes));
        if ((size_t)(scratch - reinterpret_cast<uint8_t*>(entA_)) > (size_t)kWLen * 8) throw std::runtime_error("repair lists do not fit the sort buffer");
        a.k1 = k1_; a.tbits = tbits;

        a.near = getenv("ORZ_FAST_NEAR") ? (uint32_t)atoi(getenv("ORZ_FAST_NEAR")) : 16;
        // (in a tile's first round the walk reaches over the three tiles that are still in their rounds and its answers are
        // redone in the last round anyway: one trip of four instead of up to four trips -- FastEval 121 -> 102 us a launch,
        // +0.02 % of output on the text workload, nothing on zeros with noise; none at all: 95 us, +0.07 %)
        a.near1 = getenv("ORZ_FAST_NEAR1") ? (uint32_t)atoi(getenv("ORZ_FAST_NEAR1")) : std::min<uint32_t>(a.near, 4);
        // (candidates beyond the reference's depth: half as many again -- a third at shallow depths -- keeps the sizes centred
        // on the reference's: text, full block, -l0 / -l1 / -l2: -0.20 / 0.00 / +0.17 %; 6 MB at -l0: -0.53 / -0.36 / -0.18 %
        // with 3 / 2 / 1 more than its 5; the full depth again gives -0.6 ... -0.1 %)
        a.extra = getenv("ORZ_FAST_EXTRA") ? (uint32_t)atoi(getenv("ORZ_FAST_EXTRA")) : (a.depth + 1) / (a.depth < 10 ? 3 : 2);
        a.centry = fcentry_; a.tentry = ftentry_; a.cm = fcm_; a.cp = fcp_; a.nchg = &fctl_->chg; a.nentp = &fctl_->nent;
        be_.launch(1, FastSetNent{fctl_, nent});
        a.dbg = getenv("ORZ_FAST_DBG") ? (uint32_t)atoi(getenv("ORZ_FAST_DBG")) : 0;
        a.stats = fdiag_;
        if (a.dbg & 64) be_.memset(a.stats, 0, 32 * 8);
        // Tile size: the configured one for full blocks; short inputs take finer tiles (the step count stays small
        // anyway), and a block whose parse turns out unstable -- many items lost their source -- is redone with tiles
        // a quarter the size (match-dense, highly repetitive data; never seen on text).
        // Rounds: R per tile -- and, since round 6, a SCHEDULE PER BLOCK.  The default (256 K x 4) is what zeros with noise need (one
        // hot context whose item starts depend on each other over long distances: +0.46 % against +0.81 % at three rounds); text
        // loses nothing at three rounds (a full block, emulation: +0.045 % at 256 K x 4, +0.041 % at 256 K x 3) and little with
        // tiles half as large again (kSettledTile, orz_fast.h: the measurements).  Which kind a block is, the block BEFORE it says (its statistics came with the parse's read-back:
        // no extra wait, and the same choice on every run -- reused encoders, the emulation): text-like = at least one item per ten
        // bytes, the busiest ring context under a quarter of the items, under half a per cent of the items repaired.  A stream's
        // first block, short blocks, and a block whose parse turns out unstable under the settled schedule (it is redone) take
        // the default.  ORZ_FAST_SCHED=0: off; =<tile>x<rounds>: another settled schedule (experiments).
        uint32_t T = ftile_, R = frounds_;
        bool settled = false;
        {
            static const uint32_t tdiv = getenv("ORZ_FAST_TDIV") ? (uint32_t)atoi(getenv("ORZ_FAST_TDIV")) : 128;  // aim at this many tiles per block
            const uint32_t want = ((n / tdiv + kSub - 1) / kSub) * kSub;
            T = std::max<uint32_t>(kSub, std::min<uint32_t>(ftile_, want));
            // (a full unit is not a short input -- nor is the better part of one behind other blocks: the last block of the 100 MB
            // workload, 16.1 of 16.8 MB, ran 131 steps of 126 K tiles until round 6.  A stream's first block keeps the rule it had.)
            const bool big_part = !stream_start_ && n >= cur_unit_ / 2;
            if (n >= cur_unit_ || big_part) T = ftile_;
            // The first block of a longer stream has nothing to overlap with (later blocks parse while the previous block's
            // symbols are ranked): it can take larger tiles (ORZ_FAST_LEADMUL) -- half the steps at 2, +0.1 % on that block's
            // output for text but +1 % for zeros with noise, and 2 ms of 330 per 100 MB: off.
            static const uint32_t lead_mul = getenv("ORZ_FAST_LEADMUL") ? (uint32_t)atoi(getenv("ORZ_FAST_LEADMUL")) : 1;
            if (lead_block_ && T == ftile_ && lead_mul >= 1 && lead_mul <= 8) T = (uint32_t)std::min<uint64_t>((uint64_t)lead_mul * ftile_, kNewMax);
            // (only whole blocks hand their statistics on: 8 MiB units under the settled schedule measured the same 193 ms per
            // 100 MB as under the default and 0.13 % more output -- half as many tiles leave the third round half as much to do)
            if (sched_auto_ && settled_next_ && (n >= cur_unit_ || big_part) && T == ftile_) { T = sched_tile_; R = sched_rounds_; settled = true; }
        }
        for (int attempt = 0;; attempt++) {
            a.tile = T;
            a.rounds = R;
            if (attempt) {  // (the first attempt finds the bitmap and the list counters as the prep left them)
                be_.memset(vbits_ + nent / 64, 0, 16);
                be_.launch_waves(((size_t)nent + 63) / 64, FastSlotInitWave{epos_, nullptr, runstart_, nent, vbits_, frlen_}, 0);
                be_.memset(fccnt_, 0, (size_t)(kNumKeys + 1) * 4);
                be_.launch(nent, FastListReset{epos_, slot_keys, runstart_, nent, fccnt_});
            }
            const size_t nn = (size_t)n + 264;
            {   // the round state and the repair stage's lists, ONE launch (round 6; before: fifteen fill dispatches)
                // (ev / farv / dirty need no reset: a position's first evaluation of a parse overwrites them before they are read)
                ZeroRanges z;
                z.add(kbits_, ((size_t)n / 64 + 2) * 8);
                z.add(k1_, ((size_t)kNewMax / 4096 + 2) * 8);
                z.add(fty_, nn); z.add(fnl_, nn); z.add(fpt_, nn); z.add(fmf_, nn); z.add(fef_, nn);
                z.add(fsbits_, ((size_t)n / 64 + 8) * 8);
                z.add(fcm_, (size_t)(nsub + 2) * 256 * 4);
                z.add(fcp_, (size_t)(nsub + 2) * 256 * 4);
                z.add(&fctl_->lastflips, 4);
                z.add(fcnew_, (size_t)(kNumKeys + 1) * 4);  // (compact lists: every run starts with its history slots, FastSlotInitWave)
                if (repair_lists) {
                    z.add(tbits, tbits_bytes);
                    z.add(kdirty, kdirty_bytes);
                    z.add(fcok_, 256 * 4);  // item starts the repairs added per context (FastCokGrow)
                    z.add(frdirty_ + kDirtyWords, (size_t)kDirtyWords * 8);  // (the first pass's rd_out; later ones: FastPassBegin)
                }
                be_.launch(z.units(), z);
            }
            be_.launch(nvw / 64 + 1, V1Build{vbits_, nvw, v1_});
            be_.launch(256, FastCpInit{ctxcount_, fcp_, ftentry_});
            // ---- pipelined Gauss-Seidel rounds
            const uint32_t ntile = (n + T - 1) / T, cpt = T / kSub;
            // ring horizons of the first tile (no counts yet: the history alone)
            // (one subtile more than the tile: the first step also evaluates the two positions behind it for the lazy rules)
            be_.launch_waves(256, FastPrefix{a, 0, 0, std::min(cpt + 1, nsub), cpt, 0}, 0);
            { const FastHorizon fh{a, 0, std::min(cpt + 1, nsub) - 1}; be_.launch(fh.threads(), fh); }
            // a full block's round loop is the same launch sequence every time: replay it as a hipGraph
            static const bool fused_steps = !(getenv("ORZ_FAST_FUSED") && !strcmp(getenv("ORZ_FAST_FUSED"), "0"));  // (experiments: every kernel a launch of its own)
            const bool use_graph = be_.graphs_enabled() && (n == kNewMax || n == cur_unit_);
            const uint64_t gkey = ((uint64_t)R << 58) | ((uint64_t)T << 32) | n;
            const bool replayed = use_graph && be_.graph_replay(gkey);
            struct CaptureGuard {  // a launch that throws inside the capture must not leave the stream capturing
                BE& be;
                bool on;
                ~CaptureGuard() { if (on) be.graph_capture_abort(); }
            } capture{be_, use_graph && !replayed};
            if (capture.on) be_.graph_capture_begin();
            if (replayed) stats.sweeps += ntile + R - 1;
            for (uint32_t step = 1; step <= ntile + R - 1 && !replayed; step++) {
                const uint32_t t_lo = step > R ? step - R : 0, t_hi = std::min(step - 1, ntile - 1);
                const uint32_t lo = kPre + t_lo * T;
                const uint32_t hi = (uint32_t)std::min<uint64_t>(len, (uint64_t)kPre + (uint64_t)(t_hi + 1) * T);
                const uint32_t hi2 = std::min(len, hi + 2);
                be_.timed_begin();
                // the newest active tile is in its first round while tiles are still being started; the two positions behind
                // the range (the lazy rules look ahead) have never been evaluated either
                const uint32_t r1lo = step <= ntile ? kPre + t_hi * T : hi;
                // the tiles in their first two rounds are evaluated in full; the flips of this step mark below the next step's line
                const uint32_t r2lo = step >= 2 && step - 2 < ntile ? kPre + (step - 2) * T : (step < 2 ? kPre : hi);
                const uint32_t mark_hi = step - 1 < ntile ? kPre + (step - 1) * T : len;
                // (the compact lists hold the tiles that had their last round before this step)
                const uint32_t cline = kPre + (step > R ? step - R : 0) * T;
                // (a 64-register build of FastEval -- eight waves per SIMD instead of six, 41 registers spilled -- measured 134 vs 124 us)
                be_.launch(hi2 - lo, FastEval{a, lo, hi2, r1lo, r2lo, step, cline});
                be_.timed_end();
                const uint32_t c0 = t_lo * cpt, nc = (hi - (kPre + c0 * kSub) + kSub - 1) / kSub, nt = t_hi - t_lo + 1;
                be_.timed_begin(3);
                if (fused_steps) {
                    // ONE launch decides (FastDecide), draws up the chunk maps (PathUpWave) and folds the counters of the tile that
                    // retired in the step before (FastRetireDone): under several encoders a launch waits its turn for ~100 us
                    // whatever its size, so a step is six launches now (round 4: ten)
                    const bool had_retire = step - 1 >= R && step - 1 < ntile + R - 1;
                    const uint32_t plo = had_retire ? kPre + (step - 1 - R) * T : kPre, phi = had_retire ? (uint32_t)std::min<uint64_t>(len, (uint64_t)plo + T) : kPre;
                    PathUpWave up{a, c0};
                    up.decide = 1; up.nup = nc * 4; up.done = FastRetireDone{a, plo, phi, fcut_};
                    be_.launch_waves((size_t)nc * 4 + (phi - plo + 63) / 64, up, PathUpWave::lds_bytes());
                } else {
                    be_.launch(hi - lo, FastDecide{a, lo, hi});
                    be_.launch_waves((size_t)nc * 4, PathUpWave{a, c0}, PathUpWave::lds_bytes());
                }
                be_.timed_end(3);
                be_.launch_group(PathTileDown{a, t_lo, nt});
                be_.launch_waves(nc, PathMarkWave{a, c0}, PathMarkWave::lds_bytes());
                const uint32_t fhi = std::min(len, hi + 240);
                // The end of a step is not a chain: the ordinal prefix (extrapolated over the tile that starts next, and one subtile
                // more: the next step also evaluates the two positions behind its newest tile) needs the path's counts but not
                // the flips, the ring horizons need the prefix but not the retiring tile -- each pair is ONE grid (orz_fast.h),
                // so that they run side by side whatever queues the runtime hands out.  The tile that has just had its last round
                // is final: its item starts join the compact lists (while a later tile will still read them).
                const FastFlip ff{a, lo, fhi, t_hi + 1, mark_hi, step >= R ? (uint32_t)std::min<uint64_t>(len, (uint64_t)lo + T) : 0, &fctl_->lastflips};
                const uint32_t ext = std::min(cpt + 1, nsub - std::min(nsub, c0 + nc));
                const FastPrefix fp{a, c0, c0 + nc, ext, cpt, step >= R ? std::min(c0 + cpt, c0 + nc) : c0};
                const FastHorizon fh{a, c0, c0 + nc + ext - 1};
                const bool retire = step >= R && step < ntile + R - 1;
                const uint32_t rlo = retire ? kPre + (step - R) * T : kPre, rhi = retire ? (uint32_t)std::min<uint64_t>(len, (uint64_t)rlo + T) : kPre;
                if (fused_steps) {
                    const uint32_t nflip = fhi - lo + 1, nret = rhi - rlo;
                    be_.launch_waves(flip_blocks(nflip) + 256, FlipPrefixWave{ff, fp, nflip}, FlipPrefixWave::lds_bytes());
                    be_.launch_waves((size_t)(nret + 63) / 64 + (fh.threads() + 63) / 64, RetireHorizonWave{FastRetire{a, rlo, rhi, fcut_}, fh, nret}, 0);
                    // (FastRetireDone: in the next step's FastDecide grid -- every retiring step is followed by one)
                } else {
                    be_.launch((size_t)fhi - lo + 1, ff);
                    be_.launch_waves(256, fp, 0);
                    be_.launch(fh.threads(), fh);
                    if (retire) {
                        be_.launch(rhi - rlo, FastRetire{a, rlo, rhi, fcut_});
                        be_.launch(rhi - rlo, FastRetireDone{a, rlo, rhi, fcut_});
                    }
                }
                stats.sweeps++;
            }
            // ---- frozen boundaries: sources, cuts, exact predictor -- until nothing changes.  The passes are launched in
            // groups without the host in between: a pass that finds nothing to repair sets `done` on the device and the
            // kernels of later passes return at once; the control block is read once per group.
            const bool incr_repairs = !getenv("ORZ_FAST_FULLPASS");  // (tests, experiments: every pass walks for every match)
            // (a walk is ONE thread's chain of dependent loads, and a launch lasts as long as its longest walk: 256 item starts ->
            // 258 us a launch of FastSourceL on the text workload, 64 -> 160 us for +0.02 % of output, 32 -> 147 us, +0.07 %, 16: +0.4 %)
            // The cap follows the level's depth (24 / 64 / 184 at -l0 / -l1 / -l2): with 64 at -l2 the text workload came out +0.28 %.
            const uint32_t src_cap = getenv("ORZ_FAST_SRCCAP") ? (uint32_t)atoi(getenv("ORZ_FAST_SRCCAP")) : 4 * a.depth + 4;  // (0 = no limit)
            static const bool ord_ballots = !(getenv("ORZ_FAST_ORD") && !strcmp(getenv("ORZ_FAST_ORD"), "table"));  // (experiments: the LDS-table form)
            const FastFlip flip_all{a, kPre, len, ~0u, 0, 0, &fctl_->lastflips};
            const uint32_t tw = n / 64 + 1;  // words of tbits that can hold a bit (positions kPre .. len)
            int pass = 0;
            auto lists_pass = [&]() {
                    // the bitmaps in slot order follow the path: everywhere before the first pass, from then on at the positions
                    // the repair kernels rewrote
                    if (pass == 0) be_.launch((size_t)n + 1, flip_all);
                    else be_.launch(tw, FastFlipSparse{flip_all, tw, kdirty, fctl_});
                    // exact ordinals of the item starts (per-(subtile, ctx) counts, their prefix, rank inside the subtile) and the
                    // subtiles' lists of matches and WORD items
                    uint64_t* rd_in = frdirty_ + (size_t)(pass & 1) * kDirtyWords;
                    uint64_t* rd_out = frdirty_ + (size_t)((pass + 1) & 1) * kDirtyWords;
                    // (which contexts have grown by more than the edge margin since the first pass: from the counters the rewrite
                    // kernels keep -- the lists below are drawn up before this pass's ordinals exist)
                    {   // (+ the clearing of this pass's run flags; the first pass's were cleared with the round state)
                        FastCokGrow cg{fcok_, fcok_ + 256};
                        cg.ctl = fctl_;
                        if (pass) { cg.rd_out = rd_out; cg.nwords = kDirtyWords; }
                        be_.launch(cg.threads(), cg);
                    }
                    be_.launch_waves(nsub, RepairListWave{a, mlist, wlist, mcnt, wcnt, pass && incr_repairs ? rd_in : nullptr, fdirty_, fcok_ + 256, fctl_},
                                     RepairListWave::lds_bytes());
                    col_scan(fcm_, nsub, fcp_, fctl_);
                    be_.launch(256, FastItemTotal{fcp_, nsub, fctl_});
                    if (ord_ballots) be_.launch_waves(nsub, OrdWave2{win, fsbits_, fcp_, n, ORD_, fctl_}, OrdWave2::lds_bytes());
                    else be_.launch_waves(nsub, OrdWave{win, fsbits_, fcp_, n, ORD_, fctl_}, OrdWave::lds_bytes());
                    FastSource fs{a, SRC_, fcut_, pass && incr_repairs ? rd_in : nullptr, src_cap, fctl_, fdirty_, fcok_ + 256};
                    fs.cutlist = cutlist; fs.ncut = &fctl_->ncut;
                    const uint32_t per = kListThreads;  // (the first pass with a thread per list slot -- 1024 a subtile, no loop: 138 -> 197 us a launch)
                    be_.launch((size_t)nsub * per, FastSourceL{fs, mlist, mcnt, nsub, per});
                    FastRecut rc{a, fcut_, rd_out};
                    rc.wextra = wextra; rc.nwx = &fctl_->nwx; rc.cgrow = fcok_;
                    be_.launch(kRepairGrid, FastRecutL{rc, cutlist, fctl_, kRepairGrid});
                    be_.launch(tw, FastFlipSparse{flip_all, tw, kdirty, fctl_});
                    if (pass == 0) {  // every WORD item against the running maximum of the update bits; later passes search (FastWordCheckL)
                        be_.launch(nk, KbitVals{kbits_, nk, f32_});
                        be_.inclusive_max_scan_u32(f32_, flaste_, nk);
                    }
                    be_.launch((size_t)nsub * kListThreads, FastWordCheckL{a, pass == 0 ? flaste_ : nullptr, kdirty, wextra, fixlist, fctl_, wlist, wcnt, nsub});
                    be_.launch(kRepairGrid, FastWordApplyL{a, fixlist, rd_out, fcok_, fctl_, kRepairGrid, kdirty});
                    be_.launch(1, FastPassEnd{fctl_});
            };
            // ... and what follows a group of passes in the lists form: the running maximum of the FINAL update bits (FastCommit,
            // FastWordsCarry), the per-position arrays of the post stage (idempotent: a block that turns out not to be done
            // commits again) and the item count
            auto lists_tail = [&]() {
                be_.launch(nk, KbitVals{kbits_, nk, f32_});
                be_.inclusive_max_scan_u32(f32_, flaste_, nk);
                be_.launch(n, FastCommit{a, flaste_, &fctl_->lt, S_, TY_, ML_, W0_});
                be_.launch(n, Flags32{S_, n, f32_});
                be_.exclusive_scan_u32(f32_, sc32_, n);
                be_.launch(1, SumLast{sc32_, f32_, n - 1, tailkey_ + 3});
            };
            // Lists form: the first group of passes, the commit and the item count are queued behind the rounds WITHOUT a read-back
            // -- and, for a full block, inside the same captured graph: one graph launch per block for rounds + repairs + commit --,
            // then the control block is read together with the count, ONE wait for both (round 5; before: one after the sixth
            // pass, one for the count).  A block that is not done after kFirstPasses passes (rare) goes on in groups of two with
            // a read-back each, and its commit and count are redone.
            if (repair_lists && !replayed) {
                be_.launch(1, FastCtlReset{fctl_});
                for (int k = 0; k < kFirstPasses; k++, pass++) lists_pass();
                lists_tail();
            }
            capture.on = false;
            if (use_graph && !replayed) be_.graph_capture_end(gkey);
            debug_dump("r", n, nullptr);
            // The parse is queued (≈ 25 ms of device work) and the host would only wait for it at the read-back below: the
            // output of the block that used the NEXT tail set two blocks ago is fetched now -- its copies run on the copy
            // stream beside the parse instead of between this block's parse and its item stage, where the main stream stood
            // idle for them.
            // (chunk_ends = nullptr is right here: callers that ask for chunk ends -- the object-level API -- drain every block at
            // once (post_stage's last line), so they never have a block pending at this point)
            if (ts_[cur_set_].pending && !pend_order_.empty() && pend_order_.front() == cur_set_) collect_one(out, nullptr);
            FastCtl h{};
            if (repair_lists) {
                pass = kFirstPasses;
                for (int group = 0; group < 64; group++) {
                    be_.d2h_async(&h, fctl_, sizeof h);
                    be_.d2h_async(pre_two_, tailkey_ + 3, 8);
                    be_.sync();
                    pre_items_valid_ = h.done != 0;
                    if (h.done) break;
                    for (int k = 0; k < 2; k++, pass++) lists_pass();
                    lists_tail();
                }
            } else {
                be_.launch(1, FastCtlReset{fctl_});
                for (int group = 0; group < 64 && !h.done; group++) {
                    const int todo = group == 0 ? 6 : 2;  // (text: five passes that repair something and one that finds nothing)
                    for (int k = 0; k < todo; k++, pass++) {
                    be_.launch((size_t)n + 1, FastFlip{a, kPre, len, ~0u, 0, 0, &fctl_->lastflips});
                    // exact ordinals of the item starts: per-(subtile, ctx) counts, their prefix, rank inside the subtile
                    be_.launch_waves(nsub, CountWave{a, 0}, CountWave::lds_bytes());
                    col_scan(fcm_, nsub, fcp_);
                    be_.launch(256, FastItemTotal{fcp_, nsub, fctl_});
                    be_.launch_waves(nsub, OrdWave{win, fsbits_, fcp_, n, ORD_, fctl_}, OrdWave::lds_bytes());
                    // the first pass walks for every match; later ones only where the previous pass added item starts
                    uint64_t* rd_in = frdirty_ + (size_t)(pass & 1) * kDirtyWords;
                    uint64_t* rd_out = frdirty_ + (size_t)((pass + 1) & 1) * kDirtyWords;
                    be_.memset(rd_out, 0, (size_t)kDirtyWords * 8);
                    be_.launch(256, FastCtxOk{fcp_, nsub, fcok_, fcok_ + 256, pass == 0, fctl_});
                    // (the round loop's dirty flags are dead by now: their array holds the sources' ring-edge flags)
                    be_.launch(n, FastSource{a, SRC_, fcut_, pass && incr_repairs ? rd_in : nullptr, src_cap, fctl_, fdirty_, fcok_ + 256});
                    be_.launch(n, FastRecut{a, fcut_, rd_out});
                    be_.launch((size_t)n + 1, FastFlip{a, kPre, len, ~0u, 0, 0, &fctl_->lastflips});
                    be_.launch(nk, KbitVals{kbits_, nk, f32_});
                    be_.inclusive_max_scan_u32(f32_, flaste_, nk);
#if defined(ORZ_RACE_SELFTEST)
                    be_.launch(n, FastWordCheckRacy{a, flaste_, rd_out, fctl_});
#else
                    be_.launch(n, FastWordCheck{a, flaste_, fcut_, fctl_});
                    be_.launch(n, FastWordApply{a, fcut_, rd_out});
#endif
                    be_.launch(1, FastPassEnd{fctl_});
                    }
                    be_.d2h(&h, fctl_, sizeof h);
                }
            }
            if (!h.done) throw std::runtime_error("fast parse: repairs did not converge");
            hfin_ = h;
            if (getenv("ORZ_FAST_SHOWFLIPS")) fprintf(stderr, "T=%u: %u item starts changed in their last round, %u repairs in %u passes, %u items\n", T, h.lastflips, h.total, h.passes, h.nmem);
            stats.seg_evals += h.total;  // (fast mode: repairs made)
            // unstable = more than 1 % of the items repaired AND more than one repair per 2000 input bytes (sparse item
            // streams -- long zero runs -- reach the first mark with a handful of repairs that cost nothing).  Match-dense
            // synthetic text (tests' "mixed" shape, 8 MB): one parse +0.49 % vs the oracle, redone at 64 K tiles below it.
            static const uint32_t redo_div = getenv("ORZ_FAST_REDO_DIV") ? (uint32_t)std::max(1, atoi(getenv("ORZ_FAST_REDO_DIV"))) : 100;
            if (T <= kSub || (uint64_t)h.total * redo_div < (uint64_t)h.nmem || (uint64_t)h.total * 2000 < (uint64_t)n) break;
            if (settled) { T = ftile_; R = frounds_; settled = false; }  // (the settled schedule did not suit this block: the default first)
            else T = std::max<uint32_t>(kSub, (T / 4 + kSub - 1) / kSub * kSub);
            stats.seg_evals -= h.total;  // (count the repairs of the parse that is kept)
        }
        // the next block's schedule, from this block's statistics (see above)
        settled_next_ = n == kNewMax && (uint64_t)hfin_.nmem * 10 >= n && (uint64_t)hfin_.hot * 4 <= hfin_.nmem && (uint64_t)hfin_.total * 200 < hfin_.nmem;
        if (getenv("ORZ_FAST_SHOWSCHED")) fprintf(stderr, "block %llu: T=%u R=%u%s; items %u, busiest context %u, repairs %u -> next block %s\n", (unsigned long long)stats.blocks, T, R,
                                                  settled ? " (settled)" : "", hfin_.nmem, hfin_.hot, hfin_.total, settled_next_ ? "settled" : "default");
        if (a.dbg & 64) {  // diagnostics: counters of the flips
            unsigned long long h[32];
            be_.d2h(h, a.stats, sizeof h);
            fprintf(stderr, "flip: %llu item flips, %llu word flips, walk trips %llu / %llu\n", h[16], h[17], h[18], h[19]);
        }
        // Every match of the frozen parse against first principles (FastVerify: the source is an item start of the same
        // context, inside the ring, with the same bytes).  On by default since the round-3 incident: the
        // counters stay on the device until the stream ends (no synchronisation) and a finding fails the encode -- no stream
        // is better than one no decoder accepts.  ORZ_FAST_VERIFY=0 off, 1 = read back and reported per block.
        if (const int vmode = verify_mode()) {
            unsigned long long* e = fdiag_ + 32;
            if (vmode != 2 || stream_start_) be_.memset(e, 0, 6 * 8);
            be_.launch(n, FastVerify{a, SRC_, S_, e});
            if (vmode != 2) report_verify("block");
        }
        // ---- hand over to the post stage; carry the model state (the last pass changed nothing: its counts are final)
        if (!repair_lists) be_.launch(n, FastCommit{a, flaste_, &fctl_->lt, S_, TY_, ML_, W0_});  // (lists form: committed and counted with the last read-back)
        be_.launch(1, FastLtCarry{fpt_, n, fctl_});
        be_.launch(32768, FastWordsCarry{a, flaste_, krunend_, wsnap_});
        be_.launch(256, FastCtxCarry{fcp_, nsub, ctxcount_});
        debug_dump("f", n, &hfin_);
    }

    // Second half of a block: items -> len_min -> symbols -> symrank -> Huffman -> bit pack (shared by both parse modes).
    // Three streams: the item stage and the per-context gather run on the main stream behind the parse; the symbol-ranking
    // launches of consecutive blocks follow each other on stream 1 (the per-context chains never reset, so that stream IS
    // the serial chain of a stream: nothing else sits on it); histograms, Huffman and bit packing of block k run on stream 2
    // beside the ranking of block k+1.  Blocks take the two buffer sets alternately: block k+2 reuses the set of block k
    // after its output was collected.
    template <class OutT>
    void post_stage(uint32_t n, uint32_t len, OutT& out, std::vector<size_t>* chunk_ends, double t2) {
        const uint8_t* win = dwin();
        const int b = cur_set_;
        cur_set_ ^= 1;
        TailSet& t = ts_[b];
        // ---- the block that used this set two blocks ago has long finished: take its output (in block order)
        if (t.pending) collect_one(out, nullptr);
        // (device output: the frame kernels of the block that used this set before run on the copy stream -- this block's stages
        // must not overwrite the chunk buffers and flags they read; the side streams follow the main stream's events)
        if (t.framed) { be_.wait(kEvOut + b); t.framed = false; }
        // ---- items
        uint32_t nitems = 0;
        {
            uint32_t two[2] = {0, 0};
            if (pre_items_valid_) {  // (fast parse, lists form: flags, scan and count came with the repairs' read-back)
                two[0] = pre_two_[0]; two[1] = pre_two_[1];
                pre_items_valid_ = false;
            } else {
                be_.launch(n, Flags32{S_, n, f32_});
                be_.exclusive_scan_u32(f32_, sc32_, n);
                be_.launch(1, SumLast{sc32_, f32_, n - 1, tailkey_ + 3});
                be_.d2h(two, tailkey_ + 3, 8);
            }
            nitems = two[0];
            hist_hint_ = n == kNewMax && nitems >= 1 ? nitems - 1 - two[1] : ~0u;  // (valid for a slide by the whole block: slide_by.
            // Tried in round 6 for the 8 MiB units too -- the unit's items plus the unit's before it, less the two positions that
            // leave: the arithmetic holds (ORZ_CHECK_HINTS) but the stream came out 1 % slower: the read-back paces the host)
        }
        if (nitems > t.cap) grow_tail_set(t, nitems);  // (the set is idle: its last block was collected above)
        be_.launch(n, CompactPos32{f32_, sc32_, n, kPre, t.ipos});
        if (inject_kind_ && inject_kind_ != kViLenMin2) be_.launch(1, VerInjectK{inject_kind_, inject_nth_, t.ipos, nitems, TY_, ML_, SRC_, ORD_, win, S_});  // (tests of the gate)
        // len_min of each reference (keys reuse the sort buffers)
        be_.launch(nitems, LenMinKeys{t.ipos, TY_, SRC_, nitems, entA_});
        const uint64_t* lk = be_.sort_u64(entA_, entB_, nitems, 2 * kPosBits, kPosBits);  // (the keys come in position order: a stable sort by source)
        be_.launch(nitems, LenMinEval{lk, nitems, ML_, LENMIN_, LMV_});
        be_.launch(nitems, LenMinCommit{lk, nitems, ML_, LMV_, LENMIN_});
        if (inject_kind_ == kViLenMin2) be_.launch(1, VerInjectLmv{inject_nth_, t.ipos, nitems, TY_, LMV_});  // (tests of the gate)
        be_.launch(nitems, ItemSyms{win, t.ipos, nitems, TY_, ML_, W0_, LMV_, SRC_, ORD_, t.isym, t.ictx, t.iunl, t.ienc, t.irob,
                                    t.ial});
        const uint32_t nchunks = (nitems + kChunkItems - 1) / kChunkItems;
        if (nchunks > kMaxChunks) throw std::runtime_error("too many chunks in a block");
        if (stream_start_) {  // src/lz.rs:238-265
            be_.memset(counts_, 0, (kSyms + 3) * 4);
            be_.launch_waves(((size_t)std::min(nitems, kChunkItems) + 4095) / 4096, CensusCountWave{t.isym, std::min(nitems, kChunkItems), counts_},
                             CensusCountWave::lds_bytes());
            be_.launch(kSyms, CensusOrder{counts_, order_, ncounted_});
            be_.launch((size_t)512 * kSyms, CensusFill{order_, srstate_});
        }
        // ---- model state carried to the next block (still on the main stream)
        if (!fast_) {
            const uint32_t nseg = (n + seg_ - 1) / seg_;
            be_.d2d(ctxcount_, base_ + (size_t)(nseg % ring_) * 256, 256 * 4);
            uint64_t ex;
            be_.d2h(&ex, exitst_ + nseg, 8);
            const uint8_t ltf = (uint8_t)(ExitPair::exit(ex) & 3);
            be_.launch(32768, WordsLastRun{kbits_, k1_, k2_, kpos_, krun_, krunend_, wlast_});
            be_.launch(32768, WordsApply{win, wlast_, len, (uint32_t)ltf, wsnap_});
            lt_carry_ = ltf;
        }  // (the fast mode carried its state at the end of fast_parse)
        // ---- the items of each context side by side (stable sort by context: 9 key bits), still on the main stream
        be_.sort_by_ctx(t.ictx, t.skey, t.sperm, nitems);
        be_.launch(nitems, SymGather{t.sperm, t.isym, t.iunl, nitems, t.gsym});
        be_.launch(513, SymRunStart{t.skey, nitems, t.rstart});
        be_.record(kEvItems + b);  // items of this block are ready
        release_token();           // (prep, parse and item stage are queued: the next encoder may start its block)
        // ---- symbol ranking: 512 independent serial chains, launch after launch on stream 1
        MainStreamGuard back_to_main{be_};
        be_.select(1);
        be_.wait(kEvItems + b);
        be_.symrank(srstate_, t.gsym, t.grank, t.rstart, nitems, t.srflags, srbackup_, t.skey);  // (skey: free since SymRunStart)
        be_.record(kEvRank + b);
        // ---- static Huffman per chunk and bit packing on stream 2
        be_.select(2);
        be_.wait(kEvRank + b);
        be_.launch(nitems, SymScatter{t.sperm, t.grank, nitems, t.irank});
        be_.memset(t.hw, 0, (size_t)nchunks * kHwStride * 4);
        be_.launch_waves(((size_t)nitems + 4095) / 4096, HistWave{t.irank, t.ial, t.ienc, nitems, t.hw}, HistWave::lds_bytes());
        be_.huffbuild(HuffBuild{t.hw, nchunks, t.hl, t.hc});
        be_.launch(nitems, ItemBits{t.irank, t.ial, t.ienc, t.irob, t.hl, nitems, t.blen});
        be_.exclusive_scan_u32(t.blen, t.bscan, nitems);
        be_.memset(t.out, 0, (size_t)nchunks * kChunkCapWords * 4);
        be_.launch_waves(nchunks, ChunkHeaderWave{t.hl, nchunks, nitems, len, unit_base_, t.ipos, order_, ncounted_, stream_start_ ? 1 : 0,
                                                  t.out, outoff_, t.hdrbits}, ChunkHeaderWave::lds_bytes());
        be_.launch(nitems, Pack{t.irank, t.ial, t.ienc, t.irob, t.hl, t.hc, t.bscan, t.hdrbits, outoff_, nitems, t.out});
        be_.launch(nchunks, ChunkTotals{t.bscan, t.blen, t.hdrbits, nitems, nchunks, t.tot});
        be_.record(kEvTail + b);  // this block's bytes are ready
        be_.select(0);
        // ---- the validity gate (orz_verify.h): the items as they will be coded against the decoder's rules, with state of its
        // own; on the main stream BEHIND the hand-over to the ranking chain (it only reads the items), its findings are read
        // with the block's output and fail the encode before a byte of the block is handed out
        uint32_t* verr = t.srflags + 4;
        if (gate_on()) {
            VerArgs v;
            v.win = win; v.ipos = t.ipos; v.isym = t.isym; v.ictx = t.ictx; v.irob = t.irob; v.iunl = t.iunl; v.ienc = t.ienc; v.ial = t.ial;
            v.nitems = nitems; v.end = len; v.ML = ML_; v.SRC = SRC_; v.ORD = ORD_; v.sperm = t.sperm; v.rstart = t.rstart;
            v.vrec = vrec_; v.vord = vord_; v.vctx = vctx_; v.vlast = vlast_; v.vwords = vwords_; v.err = verr;
            be_.launch(kVeCount, VerInit{verr});
            { ZeroRanges z; z.add(vrec_ + kPre, (size_t)n * 4); be_.launch(z.units(), z); }
            // three launches and the two sorts (round 6; before: nine launches) -- the keys of both sorts are built by the first,
            // each in its half of the sort buffers (at most 2^24 items: half of a buffer's 2^25 entries)
            uint64_t* lmA = entA_;
            uint64_t* lmB = entB_;
            uint64_t* evA = entA_ + kWLen / 2;
            uint64_t* evB = entB_ + kWLen / 2;
            be_.launch(nitems, VerStage1{v, lmA, evA});
            const uint64_t* lmk = be_.sort_u64(lmA, lmB, nitems, 2 * kPosBits, kPosBits);  // (the keys come in item order)
            const uint64_t* evs = be_.sort_u64(evA, evB, nitems, 2 * kPosBits - 9, kPosBits);  // 15 key bits + the sentinel's bit above 25 position bits that come in order
            be_.launch(nitems, VerStage2{v, lmk, evs});
            { const VerStage3 s3{v, lmk, evs}; be_.launch(s3.threads(), s3); }
        } else {
            be_.launch(kVeCount, VerInit{verr});
        }
        be_.record(kEvGate + b);
        t.pending = true;
        t.nitems = nitems; t.nchunks = nchunks; t.len = len; t.block = (uint32_t)stats.blocks;
        pend_order_.push_back(b);
        stream_start_ = false;
        stats.t_post += be_.now() - t2;
        stats.blocks++;
        stats.items += nitems;
        stats.chunks += nchunks;
        stats.in_bytes += n;
        if (chunk_ends || trace) collect(out, chunk_ends);  // callers that need this block's bytes now
    }

    // Append the output of every block whose tail stage is in flight, oldest first.
    template <class OutT>
    void collect(OutT& out, std::vector<size_t>* chunk_ends) {
        while (!pend_order_.empty()) collect_one(out, pend_order_.size() == 1 ? chunk_ends : nullptr);
    }
    // ... of the oldest one
    template <class OutT>
    void collect_one(OutT& out, std::vector<size_t>* chunk_ends) {
        if (pend_order_.empty()) return;
        TailSet& t = ts_[pend_order_.front()];
        pend_order_.erase(pend_order_.begin());
        t.pending = false;
        const uint32_t nitems = t.nitems, nchunks = t.nchunks, len = t.len;
        // The copies run on a stream of their own behind THIS block's tail stage: on the tail stream they would queue behind
        // the next block's tail, which waits for that block's symbol ranking -- and the host with them (3.3 ms a block of
        // the ranking chain standing idle, measured).
        MainStreamGuard back_to_main{be_};
        const int set = (int)(&t - ts_);
        if (dev_out_) {
            // the stream stays in device memory: the device frames the block behind its tail stage and gate, no host wait, no read-back
            be_.select(3);
            be_.wait(kEvTail + set);
            be_.wait(kEvGate + set);
            const FrameLayout lay{t.tot, t.srflags, nchunks, (uint32_t)kChunkCapWords};
            be_.launch(kFrameThreads, FrameChunks{lay, t.out, octl_, dout_, kFrameThreads});
            be_.launch(1, FrameAdvance{lay, octl_, t.block});
            if (out_inject_ && t.block == 0 && nchunks) be_.launch(1, FrameInject{dout_, (unsigned long long)out_inject_});  // (tests of ORZ_VERIFY=decode)
            be_.record(kEvOut + set);
            t.framed = true;
            if (chunk_ends) {  // end_spos of each chunk, src/lz.rs:268 (callers that ask for them read every block at once)
                for (uint32_t i = 0; i < nchunks; i++) {
                    uint32_t e = len;
                    const uint32_t i1 = (i + 1) << 20;
                    if (i1 < nitems) be_.d2h(&e, t.ipos + i1, 4);
                    chunk_ends->push_back(e);
                }
            }
            collect_trace(t, nitems, len);
            be_.select(0);
            return;
        }
        if (getenv("ORZ_COPY_STREAM") && atoi(getenv("ORZ_COPY_STREAM")) == 0) be_.select(2);  // (experiments)
        else { be_.select(3); be_.wait(kEvTail + set); }
        be_.wait(kEvGate + set);
        // Two waits per collected block (round 5; before: one per size read, per flag read and per chunk): sizes and guard
        // flags together, then every chunk's bytes together -- the output buffer is grown ONCE before the copies are queued (a
        // copy in flight must not lose its target to a reallocation).
        std::vector<uint32_t> tot(nchunks);
        uint32_t f[4 + kVeCount];
        be_.d2h_async(tot.data(), t.tot, nchunks * 4);
        be_.d2h_async(f, t.srflags, sizeof f);
        be_.sync();
        {   // the guard of the block's symbol ranking (backend symrank): repeated? still impossible ranks?
            report_gate(f + 4, t.block);
            if (f[2]) fprintf(stderr, "orz: two runs of the symbol ranking of block %u from the same tables differ in %u ranks\n", t.block, f[2]);
            if (f[0]) {
                stats.rank_redos++;
                fprintf(stderr, "orz: the symbol ranking of block %u was repeated (%u impossible ranks in its first run; %u after the second)\n", t.block, f[0], f[1]);
            }
            if (f[1]) throw std::runtime_error("symbol ranking produced impossible ranks twice: the encode fails, nothing of this block is handed out");
        }
        std::vector<size_t> at_of(nchunks), tb_of(nchunks);
        size_t need = out.size();
        for (uint32_t i = 0; i < nchunks; i++) {
            const size_t tb = ((size_t)tot[i] + 31) / 32 * 4;  // finish pads to 32 bits, src/coder.rs:75-82
            if (tb / 4 > kChunkCapWords) throw std::runtime_error("chunk payload overflow");
            tb_of[i] = tb;
            size_t v = tb, lenb = 1;
            while (v >= 128) { lenb++; v /= 128; }
            need += lenb + tb;
        }
        {
            const size_t keep = out.size();
            out.resize(need);   // (one growth)
            out.resize(keep);
        }
        std::vector<uint32_t> ends(chunk_ends ? nchunks : 0, len);
        for (uint32_t i = 0; i < nchunks; i++) {
            const size_t tb = tb_of[i];
            size_t v = tb;  // write_len, src/ioutil.rs:79-88
            while (v >= 128) { out.push_back((uint8_t)(128 + v % 128)); v /= 128; }
            out.push_back((uint8_t)v);
            at_of[i] = out.size();
            out.resize(at_of[i] + tb);
            be_.d2h_async(out.data() + at_of[i], t.out + (uint64_t)i * kChunkCapWords, tb);
            if (chunk_ends) {  // end_spos of the chunk, src/lz.rs:268
                const uint32_t i1 = (i + 1) << 20;
                if (i1 < nitems) be_.d2h_async(&ends[i], t.ipos + i1, 4);
            }
        }
        be_.sync();
        if (out_inject_ && t.block == 0 && nchunks && out_inject_ < tb_of[0]) out.data()[at_of[0] + out_inject_] ^= 1;  // (tests of ORZ_VERIFY=decode)
        if (chunk_ends)
            for (uint32_t i = 0; i < nchunks; i++) chunk_ends->push_back(ends[i]);
        collect_trace(t, nitems, len);
        be_.select(0);
    }
    // the optional per-item trace of a collected block (parity tests)
    template <class TS>
    void collect_trace(TS& t, uint32_t nitems, uint32_t len) {
        if (trace) {
            const size_t at = trace->pos.size();
            trace->block.resize(at + nitems, t.block);
            trace->pos.resize(at + nitems); trace->sym.resize(at + nitems); trace->ctx.resize(at + nitems);
            trace->rank.resize(at + nitems); trace->rob.resize(at + nitems); trace->unl.resize(at + nitems);
            trace->enc.resize(at + nitems); trace->al.resize(at + nitems);
            be_.d2h(trace->pos.data() + at, t.ipos, (size_t)nitems * 4);
            be_.d2h(trace->sym.data() + at, t.isym, (size_t)nitems * 2);
            be_.d2h(trace->ctx.data() + at, t.ictx, (size_t)nitems * 2);
            be_.d2h(trace->rank.data() + at, t.irank, (size_t)nitems * 2);
            be_.d2h(trace->rob.data() + at, t.irob, (size_t)nitems * 2);
            be_.d2h(trace->unl.data() + at, t.iunl, nitems);
            be_.d2h(trace->enc.data() + at, t.ienc, nitems);
            be_.d2h(trace->al.data() + at, t.ial, nitems);
            // match source and length of each item (window offsets), gathered on the host: diagnostics only
            // (callers that trace collect every block at once, so the per-position arrays still describe this block)
            std::vector<uint32_t> hsrc(len);
            std::vector<uint8_t> hml(len);
            be_.d2h(hsrc.data() + kPre, SRC_ + kPre, (size_t)(len - kPre) * 4);
            be_.d2h(hml.data() + kPre, ML_ + kPre, len - kPre);
            trace->src.resize(at + nitems); trace->mlen.resize(at + nitems);
            for (uint32_t i = 0; i < nitems; i++) {
                const uint32_t p = trace->pos[at + i];
                const bool is_match = (trace->al[at + i] & 2) != 0;
                trace->src[at + i] = is_match ? hsrc[p] : 0;
                trace->mlen[at + i] = is_match ? hml[p] : 0;
            }
        }
    }
    // ---- the stream in device memory (round 6): FrameChunks appends every block to `dbuf` (nullptr: a buffer of the encoder's own,
    // grown to `cap`) instead of the host collecting it; finish_device() closes the stream and says how long it is.  Call after
    // reset(), before the first block.
    void begin_device_output(uint8_t* dbuf, size_t cap) {
        if (!dbuf) {
            if (cap > own_cap_) {
                if (own_out_) { be_.free(own_out_); owned_.erase(std::find(owned_.begin(), owned_.end(), (void*)own_out_)); own_out_ = nullptr; own_cap_ = 0; }
                own_out_ = take<uint8_t>(cap, false);
                own_cap_ = cap;
            }
            dbuf = own_out_;
        }
        dout_ = dbuf;
        dev_out_ = true;
        MainStreamGuard back_to_main{be_};
        be_.select(3);
        be_.launch(1, FrameReset{octl_, (unsigned long long)cap});
    }
    bool device_output() const { return dev_out_; }
    struct DeviceResult { const uint8_t* data; size_t len; };
    DeviceResult finish_device() {
        if (!dev_out_) throw std::runtime_error("finish_device without begin_device_output");
        ByteBuf none;
        collect(none, nullptr);  // (frames what is pending)
        {
            MainStreamGuard back_to_main{be_};
            be_.select(3);
            be_.launch(1, FrameEof{octl_, dout_});  // EOF chunk, src/lib.rs:89
            be_.mark_end();                          // (the encode's closing time stamp: the stream is complete here)
            be_.d2h(&hctl_, octl_, sizeof hctl_);    // ONE wait: the stream's length and whatever stopped it
        }
        dev_out_ = false;
        stats.host_syncs += be_.take_host_syncs();
        const OutCtl& h = hctl_;
        if (h.rankdiff) fprintf(stderr, "orz: two runs of the symbol ranking from the same tables differ in %u ranks\n", h.rankdiff);
        if (h.redo) {
            stats.rank_redos += h.redo;
            fprintf(stderr, "orz: the symbol ranking of %u block(s) was repeated by its guard\n", h.redo);
        }
        if (h.fail == kOutGate) report_gate(h.gate, h.fail_block);
        if (h.fail == kOutRank) {
            fprintf(stderr, "orz: the symbol ranking of block %u was repeated (%u impossible ranks in its first run; %u after the second)\n", h.fail_block, h.rank_bad[0], h.rank_bad[1]);
            throw std::runtime_error("symbol ranking produced impossible ranks twice: the encode fails, nothing of this block is handed out");
        }
        if (h.fail == kOutFull) throw std::runtime_error("the output buffer is too small for the stream (block " + std::to_string(h.fail_block) + ")");
        if (h.fail == kOutChunk) throw std::runtime_error("chunk payload overflow");
        if (fast_ && stats.blocks && verify_mode() == 2) report_verify("stream");
        stats.out_bytes = (uint64_t)h.off;
        return DeviceResult{dout_, (size_t)h.off};
    }
    template <class OutT>
    void finish(OutT& out) {
        collect(out, nullptr);
        stats.host_syncs += be_.take_host_syncs();
        if (fast_ && stats.blocks && verify_mode() == 2) report_verify("stream");
    }
    static int verify_mode() {  // FastVerify (diagnostics of the fast parse; the gate below is what guards the output)
        static const int m = getenv("ORZ_FAST_VERIFY") ? atoi(getenv("ORZ_FAST_VERIFY")) : 0;
        return m;
    }
    // the validity gate: on unless ORZ_VERIFY=0 (measurements of its cost)
    static bool gate_on() {
        static const bool on = !(getenv("ORZ_VERIFY") && !strcmp(getenv("ORZ_VERIFY"), "0"));
        return on;
    }
    void report_gate(const uint32_t* e, uint32_t block) {
        uint32_t any = 0;
        for (uint32_t c = 0; c < kVeFirst; c++) any |= e[c];
        if (!any) return;
        std::string msg = "validity gate, block " + std::to_string(block) + ": the items do not decode:";
        for (uint32_t c = 0; c < kVeFirst; c++)
            if (e[c]) msg += " " + std::to_string(e[c]) + " x " + ver_name(c) + ";";
        msg += " first at window offset " + std::to_string(e[kVeFirst] - 1) + " -- the encode fails, nothing of this block is handed out (a streaming call has written the blocks before it: its output is incomplete)";
        fprintf(stderr, "orz: %s\n", msg.c_str());
        throw std::runtime_error(msg);
    }
    // (tests) damage the n-th suitable item of every block after the parse: "hole", "context", "ring", "lenmin", "word", "bytes"
    void set_inject(uint32_t kind, uint32_t nth) { inject_kind_ = kind; inject_nth_ = nth; }
    void report_verify(const char* what) {
        unsigned long long h6[6];
        be_.d2h(h6, fdiag_ + 32, sizeof h6);
        if (h6[1] | h6[2] | h6[3] | h6[4]) {
            char msg[320];
            snprintf(msg, sizeof msg, "fast parse verification, %s (%llu blocks so far): of %llu matches %llu have a source that is no item start, %llu one of another context, %llu one outside the ring, %llu one with other bytes (e.g. at window offset %llu)",
                     what, (unsigned long long)stats.blocks, h6[0], h6[1], h6[2], h6[3], h6[4], h6[5]);
            fprintf(stderr, "orz: %s\n", msg);
            throw std::runtime_error(msg);
        }
    }

    // window slide + LZEncoder::forward (src/lib.rs:83-84, src/lz.rs:82-87, src/matcher.rs:82-87):
    // the last kPre bytes move to offset 0, every position is rebased by 2^24, position 0 dies.
    // `slide_window` false = the caller re-uploads the whole window itself (object-level API).
    // When the fast mode encodes a block in units (encode_block_units) it slides by one unit at a time: after the
    // block's last unit the slides add up to the reference's 2^24.
    void slide(bool slide_window = true) { slide_by(last_n_, 0, slide_window); }
    // slide by `sh` positions; `extra` bytes behind the encoded region (the block's later units) move along
    void slide_by(uint32_t sh, uint32_t extra, bool slide_window = true) {
        if (sh != kNewMax) hist_hint_ = ~0u;
        be_.launch(3, TailKeys{dwin(), kPre + sh, tailkey_});
        // (two bytes more than the history: the context of the item start at window offset 1 -- hash1 of offset 0 -- looks at
        // the byte before the window.  The reference files a position under the context it had when it was inserted; the
        // tables here are rebuilt from the window's bytes, and with the front sentinel's zero in that place the position
        // landed in the context without the letter-or-digit bit -- a source no decoder finds there.)
        if (slide_window) be_.d2d(dwin() - 2, dwin() + sh - 2, (size_t)kPre + extra + 2);
        else be_.d2d(dwin() - 2, dwin() + sh - 2, 2);  // (the caller uploads the window from offset 0 on: the two bytes before it are kept here)
        for (uint32_t off = 0; off < kPre; off += sh) {
            be_.launch(sh, SlideArray<uint8_t>{S_, off, sh, kPre});
            be_.launch(sh, SlideArray<uint8_t>{ML_, off, sh, kPre});
            be_.launch(sh, SlideArray<uint32_t>{ORD_, off, sh, kPre});
            be_.launch(sh, SlideArray<uint8_t>{LENMIN_, off, sh, kPre});
            be_.launch(sh, SlideArray<uint32_t>{vrec_, off, sh, kPre});
            be_.launch(sh, SlideArray<uint32_t>{vord_, off, sh, kPre});
        }
        // (no synchronisation: whatever fills the window next is queued on this stream behind the slide)
    }
    // Encode the `take` new bytes at dwin()[kPre, kPre + take) -- one block of the stream loop (src/lib.rs:72-84).  The exact
    // mode takes them as one block, like the reference.  The fast mode can cut the block into units (ORZ_FAST_UNIT; off by
    // default, see unit_): the decoder accepts any chunk ends inside a block (it slides when the block is full,
    // src/lib.rs:119-124), so each unit closes its last chunk early, and the symbol ranking of unit k runs beside the
    // parse of unit k+1; a unit's parse sees the 16 MiB before it (the reference sees up to 16 MiB more).
    // The LEAD block of a longer stream is different: nothing overlaps its parse (later blocks parse while the block before
    // is ranked), so it is cut into small units (lead_unit_) -- ranking starts after the first unit's parse and is the
    // bottleneck from then on (a unit parses faster than it ranks); the lead block has no history, so the units' repeated
    // prep is cheap.
    template <class OutT>
    void encode_block_units(uint32_t take, bool lead, OutT& out) {
        uint32_t unit = fast_ ? unit_ : kNewMax;
        const bool lead_units = fast_ && lead && lead_unit_ && lead_unit_ < unit;
        if (lead_units) unit = lead_unit_;
        cur_unit_ = unit;
        uint32_t done = 0;
        while (done < take) {
            uint32_t n = std::min(unit, take - done);
            if (take - done - n < unit / 8) n = take - done;  // no crumbs: a short rest joins the unit before it
            set_lead_block(lead && done == 0 && !lead_units);
            unit_base_ = done;
            encode_block(n, out);
            done += n;
            if (done < take) slide_by(n, take - done);
        }
        unit_base_ = 0;
        cur_unit_ = fast_ ? unit_ : kNewMax;
    }
    void set_unit(uint32_t bytes) {  // between streams only
        if (bytes < (1u << 20) || bytes > kNewMax || bytes % kSub) throw std::runtime_error("the unit must be a multiple of 4096 in [1 MiB, 16 MiB]");
        unit_ = cur_unit_ = bytes;
    }
    void set_lead_unit(uint32_t bytes) { lead_unit_ = bytes; }  // 0 = the lead block is encoded like the others
    uint32_t lead_unit() const { return lead_unit_; }

    EncodeStats stats;
    ItemTrace* trace = nullptr;  // when set, every block appends its items

   private:
    // device allocation owned by this encoder (freed by release_all, also when the constructor throws); `zero` = false for
    // the large tables that are always written before they are read (a stream's state is ~5 GB: filling all of it costs
    // more than encoding a small input)
    template <class T>
    T* take(size_t n, bool zero = true) {
        T* p = be_.template alloc<T>(n, zero);
        owned_.push_back(p);
        return p;
    }
    void release_all() {
        for (void* p : owned_) if (p) be_.free(p);
        owned_.clear();
        be_.release_arena();
    }
    void release_token() {
        if (token_held_) { token_held_ = false; be_.parse_token_release(); }
    }
    bool token_held_ = false;
    bool pre_items_valid_ = false;   // pre_two_ holds the block's item count (and whether its second position starts an item): read
    uint32_t pre_two_[2] = {0, 0};   // with the repairs' control block
    std::vector<void*> owned_;
    BE& be_;
    Cfg cfg_;
    uint32_t seg_, wsegs_, ring_ = 0, nseg_max_ = 0, dmax_ = 0;
    bool fast_ = false;
    uint32_t ftile_ = kFastTile, frounds_ = kFastRounds, fK_ = kFastK;
    // the per-block schedule (fast_parse): on unless the caller chose tile / rounds itself
    bool sched_auto_ = false, settled_next_ = false;
    uint32_t sched_tile_ = kSettledTile, sched_rounds_ = kSettledRounds;
    bool lead_block_ = false;
    uint8_t *frows_ = nullptr, *frlen_ = nullptr, *fty_ = nullptr, *fnl_ = nullptr, *fpt_ = nullptr, *fmf_ = nullptr, *fef_ = nullptr,
            *fx0_ = nullptr, *fx1_ = nullptr, *fx2_ = nullptr, *fdirty_ = nullptr;
    uint16_t *fkw_ = nullptr, *fkmeta_ = nullptr;
    uint64_t *frdist_ = nullptr, *fwmask_ = nullptr;
    uint32_t *fhz_ = nullptr, *fhcm_ = nullptr, *fhpre_ = nullptr, *fgsum_ = nullptr;
    unsigned long long* fdiag_ = nullptr;
    uint32_t *fev_ = nullptr, *fcentry_ = nullptr, *ftentry_ = nullptr, *fcm_ = nullptr, *fcp_ = nullptr, *fcut_ = nullptr,
             *flaste_ = nullptr, *fcok_ = nullptr, *ffarv_ = nullptr;
    uint64_t *fsbits_ = nullptr, *fstext_ = nullptr, *frdirty_ = nullptr, *fcl_ = nullptr;
    uint32_t *fccnt_ = nullptr, *fcnew_ = nullptr;
    FastCtl* fctl_ = nullptr;
    FastCtl hfin_{};  // the control block as read after the last repair pass of the block parsed last (diagnostics)
    uint8_t lt_carry_ = kTyLit;
    bool stream_start_ = true;
    // buffers of the item stage and the tail stage, two sets (block parity)
    struct TailSet {
        uint32_t* ipos = nullptr;
        uint16_t *isym = nullptr, *ictx = nullptr, *irank = nullptr, *irob = nullptr, *grank = nullptr, *skey = nullptr;
        uint8_t *iunl = nullptr, *ienc = nullptr, *ial = nullptr;
        uint32_t *gsym = nullptr, *sperm = nullptr, *blen = nullptr, *bscan = nullptr, *rstart = nullptr;
        uint32_t* hw = nullptr;
        uint8_t* hl = nullptr;
        uint16_t* hc = nullptr;
        uint32_t *hdrbits = nullptr, *tot = nullptr, *out = nullptr, *srflags = nullptr;
        bool pending = false;
        bool framed = false;  // device output: the set's frame kernels were queued (the next block of the set waits for kEvOut)
        uint32_t cap = 0;  // items the per-item buffers hold (grow_tail_set)
        uint32_t nitems = 0, nchunks = 0, len = 0, block = 0;
    };
    static constexpr int kEvItems = 0, kEvRank = 2, kEvTail = 4, kEvGate = 6, kEvOut = 8, kEvIdle = 10;  // event numbers (+ set index; kEvIdle + side stream - 1)
    static constexpr uint32_t kFrameThreads = 1u << 18;  // FrameChunks' grid: 4 MiB of payload per sweep of the grid
    struct MainStreamGuard {  // whatever happens while a side stream is selected, the backend goes back to the main one
        BE& be;
        ~MainStreamGuard() { be.select(0); }
    };
    TailSet ts_[2];
    // The per-item buffers of a tail set hold `cap` items: 6 M to begin with (text has 0.3 items per byte: 5 M a block) and
    // whatever a block needs from then on, up to one item per byte -- sized for the worst case from the start they were 1.2 GB
    // of a stream's state.  Called only while the set is idle.
    static constexpr uint32_t kTailItems0 = 6u << 20;
    struct Fresh {  // buffers allocated for a growth that has not been committed yet: freed again unless `keep`
        BE& be;
        std::vector<void*> ptrs;
        bool keep = false;
        ~Fresh() {
            if (!keep) for (void* q : ptrs) if (q) be.free(q);
        }
        template <class T>
        T* get(size_t n, bool zero) {
            ptrs.reserve(ptrs.size() + 1);
            T* q = be.template alloc<T>(n, zero);
            ptrs.push_back(q);
            return q;
        }
    };
    // All-or-nothing: every new buffer is allocated BEFORE an old one is freed, so an allocation that fails --
    // eight encoders share a device -- leaves the set as it was (cap unchanged, every pointer valid) and the encode fails
    // with the allocator's error instead of a later launch on a null pointer.
    void grow_tail_set(TailSet& t, uint32_t need) {
        uint64_t cap = std::max<uint64_t>(need, (uint64_t)t.cap * 3 / 2);
        cap = std::min<uint64_t>(kNewMax, (cap + 65535) & ~65535ull);
        Fresh fresh{be_, {}};
        TailSet nt = t;
        nt.ipos = fresh.template get<uint32_t>((size_t)cap + 1, false);
        nt.isym = fresh.template get<uint16_t>(cap, true); nt.ictx = fresh.template get<uint16_t>(cap, true);
        nt.irank = fresh.template get<uint16_t>(cap, true); nt.irob = fresh.template get<uint16_t>(cap, true);
        nt.grank = fresh.template get<uint16_t>(cap, true); nt.skey = fresh.template get<uint16_t>(cap, true);
        nt.iunl = fresh.template get<uint8_t>(cap, true); nt.ienc = fresh.template get<uint8_t>(cap, true); nt.ial = fresh.template get<uint8_t>(cap, true);
        nt.gsym = fresh.template get<uint32_t>(cap, false); nt.sperm = fresh.template get<uint32_t>(cap, false);
        nt.blen = fresh.template get<uint32_t>(cap, false); nt.bscan = fresh.template get<uint32_t>(cap, false);
        owned_.reserve(owned_.size() + fresh.ptrs.size());  // (nothing below this line throws)
        fresh.keep = true;
        if (t.cap) be_.sync();
        void* old[] = {t.ipos, t.isym, t.ictx, t.irank, t.irob, t.grank, t.skey, t.iunl, t.ienc, t.ial, t.gsym, t.sperm, t.blen, t.bscan};
        for (void* q : old) {
            if (!q) continue;
            be_.free(q);
            owned_.erase(std::find(owned_.begin(), owned_.end(), q));
        }
        for (void* q : fresh.ptrs) owned_.push_back(q);
        nt.cap = (uint32_t)cap;
        t = nt;
    }
    int cur_set_ = 0;
    uint32_t last_n_ = kNewMax;  // size of the unit encoded last (what slide() slides by)
    uint32_t hist_hint_ = ~0u;   // history item starts of the next block as the host computes them (~0 = ask the device)
    uint32_t unit_base_ = 0;     // bytes of the current block encoded by earlier units (chunk headers carry decoder positions)
    // fast mode: bytes per unit of a block (ORZ_FAST_UNIT, set_unit; a multiple of 4096).  A stream of its own is a pipeline of
    // two stages -- the parse of unit k+1 beside the symbol ranking of unit k -- whose fill and drain are a unit's parse and a
    // unit's ranking.  While the ranking took 50 ms a block, units did not pay (round 4, the 100 MB workload: 8 MiB units filled
    // the pipeline 18 ms sooner but cost 28 ms of parse -- the history is sorted once per unit, the round pipeline and the repair
    // passes start once per unit --: 353 vs 350 ms).  With the ranking at 29 ms a block (round 6) they do: 16 / 10 / 8 / 6 / 4 MiB
    // units 202 / 197 / 193 / 204 / 219 ms per 100 MB, the stream 0.06 % smaller at 8 MiB.  Eight encoders on one GPU have no
    // pipeline to fill -- the GPU is busy with the others -- and pay the per-unit work: 733..770 MB/s with whole blocks, 662 with
    // 8 MiB units.  So: 8 MiB for an encoder that has the GPU to itself, the whole block for the encoders of a members job
    // (orz_capi.hip decides; the size is part of what makes a stream's bytes, so it never depends on what else is running).
    uint32_t unit_ = kNewMax / 2;
    uint32_t cur_unit_ = kNewMax;  // unit size in effect for the block being encoded
    uint32_t lead_unit_ = 0;       // unit size of a stream's lead block (0 = off)
    std::vector<int> pend_order_;  // sets whose output is still on the device, oldest first
    uint8_t* winbuf_;
    uint8_t *S_, *E_ = nullptr, *ML_, *LR_ = nullptr, *W0_, *TY_, *LENMIN_, *LMV_;
    uint32_t *ORD_, *SRC_;
    uint32_t *idx_, *kidx_, *epos_, *kpos_, *runstart_, *krun_, *krunend_;
    uint64_t *entA_, *entB_, *vbits_, *v1_, *v2_, *kbits_, *k1_, *k2_;
    SlotRec* srec_ = nullptr;
    uint64_t* exitst_ = nullptr;
    uint8_t* hist_ = nullptr;
    uint32_t* base_ = nullptr;
    ParseCtl* ctl_ = nullptr;
    uint32_t* partial_ = nullptr;
    uint32_t *f32_, *sc32_, *hpos_, *ctxcount_, *tailkey_;
    uint8_t* wsnap_;
    uint32_t* wlast_;
    uint32_t* counts_;
    uint16_t* order_;
    uint32_t* ncounted_;
    uint16_t* srstate_;
    uint32_t *vrec_ = nullptr, *vord_ = nullptr, *vctx_ = nullptr, *vlast_ = nullptr;  // the gate's decoder state (orz_verify.h)
    uint8_t* vwords_ = nullptr;
    uint32_t inject_kind_ = 0, inject_nth_ = 0;
    size_t out_inject_ = 0;  // byte of the stream's first chunk whose lowest bit is flipped on its way out (0 = none)
    uint16_t* srbackup_ = nullptr;  // the tables before the running block's ranking (the guard's second run starts from them)
    uint64_t* outoff_;
    // the stream in device memory (begin_device_output)
    bool dev_out_ = false;
    uint8_t* dout_ = nullptr;      // where the stream goes: the caller's buffer or own_out_
    uint8_t* own_out_ = nullptr;
    size_t own_cap_ = 0;
    OutCtl* octl_ = nullptr;
    OutCtl hctl_{};
};

// orz::encode (src/lib.rs:58-92) over a memory buffer that the backend can read with h2d():
// fills the window block by block, frames chunks, slides, and appends the EOF chunk.
template <class BE, class OutT>
void encode_stream(StreamEncoder<BE>& enc, BE& be, const uint8_t* src, size_t n, bool src_on_device,
                   OutT& out, bool src_pinned = false) {
    enc.reset();
    size_t off = 0;
    while (off < n) {
        uint32_t take = (uint32_t)std::min<size_t>(n - off, kNewMax);
        if (src_on_device) be.d2d(enc.dwin() + kPre, src + off, take);
        else if (src_pinned) be.h2d_pinned(enc.dwin() + kPre, src + off, take);  // async: the block's first sync covers it
        else be.h2d(enc.dwin() + kPre, src + off, take);
        enc.encode_block_units(take, off == 0 && n > enc.unit_bytes(), out);
        off += take;
        if (off < n) enc.slide();
    }
    enc.finish(out);
    out.push_back(0);  // EOF chunk, src/lib.rs:89
    enc.stats.out_bytes = out.size();
}

// The same with the finished stream left in DEVICE memory (round 6): `dbuf` = a device buffer of `cap` bytes the caller owns, or
// nullptr for one of the encoder's own (valid until the encoder's next stream).  One host wait per block (the parse's control
// block + item count) and one for the whole stream.
template <class BE>
typename StreamEncoder<BE>::DeviceResult encode_stream_device(StreamEncoder<BE>& enc, BE& be, const uint8_t* src, size_t n, bool src_on_device,
                                                              uint8_t* dbuf, size_t cap, bool src_pinned = false) {
    enc.reset();
    enc.begin_device_output(dbuf, dbuf ? cap : stream_bound(n));
    ByteBuf none;
    size_t off = 0;
    while (off < n) {
        uint32_t take = (uint32_t)std::min<size_t>(n - off, kNewMax);
        if (src_on_device) be.d2d(enc.dwin() + kPre, src + off, take);
        else if (src_pinned) be.h2d_pinned(enc.dwin() + kPre, src + off, take);
        else be.h2d(enc.dwin() + kPre, src + off, take);
        enc.encode_block_units(take, off == 0 && n > enc.unit_bytes(), none);
        off += take;
        if (off < n) enc.slide();
    }
    return enc.finish_device();
}

}  // namespace orz
