// orz_pipeline.h -- host-side orchestration of one orz stream on one device.
//
// StreamEncoder<BE> is the device-side LZEncoder (reference: /root/reference/src/lz.rs:69-346):
// it owns the window, the ring/symrank/word-predictor model state and produces, block by block,
// the framed chunks `orz::encode` (src/lib.rs:58-92) would write.  BE is a backend:
//   HipBackend  (backend_hip.hip)  -- the product: HIP kernels on gfx950
//   EmuBackend  (tests/emu)        -- host loop emulation of the same kernel bodies, tests only
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "orz_kernels.h"

namespace orz {

struct EncodeStats {
    uint64_t blocks = 0, sweeps = 0, seg_evals = 0, items = 0, chunks = 0, in_bytes = 0, out_bytes = 0;
    double t_prep = 0, t_parse = 0, t_post = 0;  // seconds (host clock around device syncs)
};

struct Flags32 {  // u32 flag per in-block position (scan input)
    const uint8_t* S;
    uint32_t n;
    uint32_t* f;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n) f[tid] = S[kPre + tid];
    }
};
struct HistFlags32 {
    const uint8_t* S;
    uint32_t* f;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < kPre) f[tid] = (tid >= 1 && S[tid]) ? 1u : 0u;
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
struct ItemPos32 {
    const uint32_t* flag;
    const uint32_t* scan;
    uint32_t n;
    uint32_t* ipos;
    ORZ_HD void operator()(size_t tid) const {
        if (tid < n && flag[tid]) ipos[scan[tid]] = kPre + (uint32_t)tid;
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

template <class BE>
class StreamEncoder {
   public:
    static constexpr size_t kChunkCapWords = (size_t)kChunkItems * 40 / 32 + 8192;  // payload words per chunk

    StreamEncoder(BE& be, Cfg cfg, uint32_t seg_size = 256, uint32_t win_segs = 0xffffffffu)
        : be_(be), cfg_(cfg), seg_(seg_size), wsegs_(win_segs) {
        if (seg_ < 256 || (seg_ & 7)) throw std::runtime_error("seg_size must be >= 256 and a multiple of 8");
        nseg_max_ = (kNewMax + seg_ - 1) / seg_;
        winbuf_ = be_.template alloc<uint8_t>((size_t)kBlock + 2 * kSent + 64);
        for (int i = 0; i < 2; i++) {
            S_[i] = be_.template alloc<uint8_t>(kWLen);
            E_[i] = be_.template alloc<uint8_t>(kWLen);
            ML_[i] = be_.template alloc<uint8_t>(kWLen);
            ORD_[i] = be_.template alloc<uint32_t>(kWLen);
            first_[i] = be_.template alloc<uint32_t>(nseg_max_ + 1);
            lt_[i] = be_.template alloc<uint8_t>(nseg_max_ + 1);
        }
        LR_ = be_.template alloc<uint16_t>(kWLen);
        SRC_ = be_.template alloc<uint32_t>(kWLen);
        W0_ = be_.template alloc<uint8_t>(kWLen);
        TY_ = be_.template alloc<uint8_t>(kWLen);
        LENMIN_ = be_.template alloc<uint8_t>(kWLen);
        LMV_ = be_.template alloc<uint8_t>(kWLen);
        idx_ = be_.template alloc<uint32_t>(kWLen);
        kidx_ = be_.template alloc<uint32_t>(kWLen);
        entA_ = be_.template alloc<uint64_t>((size_t)kWLen);
        entB_ = be_.template alloc<uint64_t>((size_t)kWLen);
        kentA_ = be_.template alloc<uint64_t>((size_t)kNewMax + 8);
        kentB_ = be_.template alloc<uint64_t>((size_t)kNewMax + 8);
        f32_ = be_.template alloc<uint32_t>(kWLen);
        sc32_ = be_.template alloc<uint32_t>(kWLen);
        hpos_ = be_.template alloc<uint32_t>(kPre + 1);
        base_ = be_.template alloc<uint32_t>(((size_t)nseg_max_ + 1) * 256);
        hist_ = be_.template alloc<uint32_t>((size_t)nseg_max_ * 256);
        csum_ = be_.template alloc<uint32_t>(((size_t)nseg_max_ / 256 + 2) * 256);
        lcnt_ = be_.template alloc<uint16_t>((size_t)nseg_max_ * 256);
        ctxcount_ = be_.template alloc<uint32_t>(256);
        fchg_ = be_.template alloc<uint32_t>(4);
        wsnap_ = be_.template alloc<uint8_t>(65536);
        wlast_ = be_.template alloc<uint32_t>(32768);
        // items
        ipos_ = be_.template alloc<uint32_t>((size_t)kNewMax + 1);
        isym_ = be_.template alloc<uint16_t>(kNewMax);
        ictx_ = be_.template alloc<uint16_t>(kNewMax);
        irank_ = be_.template alloc<uint16_t>(kNewMax);
        irob_ = be_.template alloc<uint16_t>(kNewMax);
        grank_ = be_.template alloc<uint16_t>(kNewMax);
        iunl_ = be_.template alloc<uint8_t>(kNewMax);
        ienc_ = be_.template alloc<uint8_t>(kNewMax);
        ial_ = be_.template alloc<uint8_t>(kNewMax);
        gsym_ = be_.template alloc<uint32_t>(kNewMax);
        blen_ = be_.template alloc<uint32_t>(kNewMax);
        bscan_ = be_.template alloc<uint32_t>(kNewMax);
        rstart_ = be_.template alloc<uint32_t>(520);
        counts_ = be_.template alloc<uint32_t>(kSyms + 3);
        order_ = be_.template alloc<uint16_t>(kSyms + 3);
        ncounted_ = be_.template alloc<uint32_t>(4);
        srstate_ = be_.template alloc<uint16_t>((size_t)512 * kSrWords);
        hw_ = be_.template alloc<uint32_t>((size_t)kMaxChunks * kHwStride);
        hl_ = be_.template alloc<uint8_t>((size_t)kMaxChunks * kHwStride);
        hc_ = be_.template alloc<uint16_t>((size_t)kMaxChunks * kHwStride);
        hscr_ = be_.template alloc<uint32_t>((size_t)kMaxChunks * 3 * HuffBuild::kHuffScratch);
        hdrbits_ = be_.template alloc<uint32_t>(kMaxChunks);
        tot_ = be_.template alloc<uint32_t>(kMaxChunks);
        outoff_ = be_.template alloc<uint64_t>(kMaxChunks);
        out_ = be_.template alloc<uint32_t>((size_t)kMaxChunks * kChunkCapWords);
        reset();
    }
    ~StreamEncoder() {
        void* ptrs[] = {winbuf_, S_[0], S_[1], E_[0], E_[1], ML_[0], ML_[1], ORD_[0], ORD_[1], first_[0], first_[1],
                        lt_[0], lt_[1], LR_, SRC_, W0_, TY_, LENMIN_, LMV_, idx_, kidx_, entA_, entB_, kentA_, kentB_,
                        f32_, sc32_, hpos_, base_, hist_, csum_, lcnt_, ctxcount_, fchg_, wsnap_, wlast_, ipos_, isym_,
                        ictx_, irank_, irob_, grank_, iunl_, ienc_, ial_, gsym_, blen_, bscan_, rstart_, counts_,
                        order_, ncounted_, srstate_, hw_, hl_, hc_, hscr_, hdrbits_, tot_, outoff_, out_};
        for (void* p : ptrs) be_.free(p);
    }

    // LZEncoder::new (src/lz.rs:75-80): empty rings, zero word table, after_literal = true
    void reset() {
        be_.memset(winbuf_, 0, (size_t)kBlock + 2 * kSent + 64);
        for (int i = 0; i < 2; i++) {
            be_.memset(S_[i], 0, kWLen);
            be_.memset(ML_[i], 0, kWLen);
            be_.memset(ORD_[i], 0, (size_t)kWLen * 4);
        }
        be_.memset(LENMIN_, 0, kWLen);
        be_.memset(ctxcount_, 0, 256 * 4);
        be_.memset(wsnap_, 0, 65536);
        lt_carry_ = kTyLit;
        stream_start_ = true;
        cur_ = 0;
        stats = EncodeStats();
    }

    uint8_t* dwin() { return winbuf_ + kSent; }      // device address of window offset 0
    uint8_t* dwinbuf() { return winbuf_; }           // device address of the allocation (sentinel included)

    // Encode the block whose n new bytes sit at dwin()[kPre, kPre+n).  Appends
    // { LEB128(t) chunk[t] }* (src/lib.rs:76-82, src/ioutil.rs:79-88) to `out`.
    void encode_block(uint32_t n, std::vector<uint8_t>& out) {
        if (n == 0 || n > kNewMax) throw std::runtime_error("bad block size");
        double t0 = be_.now();
        const uint8_t* win = dwin();
        const uint32_t len = kPre + n;
        const uint32_t nseg = (n + seg_ - 1) / seg_;
        // ---- history item starts -> hpos
        uint32_t nhist = 0;
        if (!stream_start_) {
            be_.launch(kPre, HistFlags32{S_[cur_], f32_});
            be_.exclusive_scan_u32(f32_, sc32_, kPre);
            uint32_t a, b2;
            be_.d2h(&a, sc32_ + (kPre - 1), 4);
            be_.d2h(&b2, f32_ + (kPre - 1), 4);
            nhist = a + b2;
            be_.launch(kPre, CompactPos32{f32_, sc32_, kPre, 0, hpos_});
        }
        // ---- sorted candidate lists
        const uint32_t nent = nhist + n;
        be_.launch(std::max<size_t>(nent, (size_t)n + 1), BuildEntries{win, hpos_, nhist, n, entA_, kentA_});
        const uint64_t* ent = be_.sort_u64(entA_, entB_, nent, 21 + kPosBits);
        const uint64_t* kent = be_.sort_u64(kentA_, kentB_, (size_t)n + 1, 15 + kPosBits);
        be_.launch(nent, ScatterIndex{ent, nent, idx_, kPre});
        be_.launch((size_t)n + 1, ScatterIndex{kent, n + 1, kidx_, kPre - 1});
        // ---- fresh speculative state for the in-block region
        for (int i = 0; i < 2; i++) {
            be_.memset(S_[i] + kPre, 0, kWLen - kPre);
            be_.memset(E_[i] + kPre, 0, kWLen - kPre);
            be_.memset(ML_[i] + kPre, 0, kWLen - kPre);
        }
        be_.memset(LENMIN_ + kPre, 0, kWLen - kPre);
        be_.launch((size_t)nseg + 1, FillFirst{first_[0], first_[1], lt_[0], lt_[1], nseg, seg_});
        be_.d2d(base_, ctxcount_, 256 * 4);
        be_.sync();
        double t1 = be_.now();
        stats.t_prep += t1 - t0;

        // ---- speculative sweeps to the causal fixed point
        uint32_t f = 0;
        while (f < nseg) {
            const uint32_t wend = (uint32_t)std::min<uint64_t>(nseg, (uint64_t)f + wsegs_);
            const int o = cur_, nw = cur_ ^ 1;
            be_.memset(fchg_, 0xff, 4);
            ParseState ps;
            ps.win = win; ps.len = len; ps.seg_size = seg_; ps.nseg = nseg;
            ps.ent = ent; ps.idx = idx_; ps.kent = kent; ps.kidx = kidx_; ps.wsnap = wsnap_;
            ps.S_old = S_[o]; ps.E_old = E_[o]; ps.ML_old = ML_[o]; ps.ORD_old = ORD_[o];
            ps.S_new = S_[nw]; ps.E_new = E_[nw]; ps.ML_new = ML_[nw]; ps.ORD_new = ORD_[nw];
            ps.first_old = first_[o]; ps.lt_old = lt_[o]; ps.first_new = first_[nw]; ps.lt_new = lt_[nw];
            ps.LR = LR_; ps.SRC = SRC_; ps.W0 = W0_; ps.TY = TY_; ps.base = base_;
            ps.first_changed = fchg_; ps.lt0 = lt_carry_; ps.cfg = cfg_;
            be_.launch(wend - f, ParseSeg{ps, f, wend, lcnt_});
            const uint32_t x0 = kPre + f * seg_;
            const uint32_t x1 = (uint32_t)std::min<uint64_t>((uint64_t)kPre + (uint64_t)wend * seg_, len);
            be_.memset(hist_ + (size_t)f * 256, 0, (size_t)(wend - f) * 256 * 4);
            be_.launch(x1 - x0, RankHist{win, S_[nw], x0, x1, seg_, hist_});
            const uint32_t nch = (wend - f + 255) / 256;
            be_.launch((size_t)nch * 256, RankChunkSum{hist_, f, wend, 256, csum_});
            be_.launch(256, RankChunkScan{csum_, base_, f, nch});
            be_.launch((size_t)nch * 256, RankApply{hist_, csum_, f, wend, 256, base_});
            be_.launch(x1 - x0, RankOrd{win, S_[nw], LR_, base_, x0, x1, seg_, ORD_[nw]});
            uint32_t fc;
            be_.d2h(&fc, fchg_, 4);
            cur_ = nw;
            stats.sweeps++;
            stats.seg_evals += wend - f;
            f = (fc == 0xffffffffu) ? wend : fc;
        }
        be_.sync();
        double t2 = be_.now();
        stats.t_parse += t2 - t1;

        // ---- items
        const int c = cur_;
        be_.launch(n, Flags32{S_[c], n, f32_});
        be_.exclusive_scan_u32(f32_, sc32_, n);
        uint32_t a, b2;
        be_.d2h(&a, sc32_ + (n - 1), 4);
        be_.d2h(&b2, f32_ + (n - 1), 4);
        const uint32_t nitems = a + b2;
        be_.launch(n, ItemPos32{f32_, sc32_, n, ipos_});
        // len_min of each reference (keys reuse the candidate buffers)
        be_.launch(nitems, LenMinKeys{ipos_, TY_, SRC_, nitems, entA_});
        const uint64_t* lk = be_.sort_u64(entA_, entB_, nitems, 2 * kPosBits);
        be_.launch(nitems, LenMinEval{lk, nitems, ML_[c], LENMIN_, LMV_});
        be_.launch(nitems, LenMinCommit{lk, nitems, ML_[c], LMV_, LENMIN_});
        be_.launch(nitems, ItemSyms{win, ipos_, nitems, TY_, ML_[c], W0_, LMV_, SRC_, ORD_[c], isym_, ictx_, iunl_,
                                    ienc_, irob_, ial_});
        const uint32_t nchunks = (nitems + kChunkItems - 1) / kChunkItems;
        if (nchunks > kMaxChunks) throw std::runtime_error("too many chunks in a block");
        if (stream_start_) {  // src/lz.rs:238-265
            be_.memset(counts_, 0, (kSyms + 3) * 4);
            be_.launch(std::min(nitems, kChunkItems), CensusCount{isym_, std::min(nitems, kChunkItems), counts_});
            be_.launch(1, CensusInit{counts_, order_, ncounted_, srstate_});
        }
        // symbol ranking: 512 independent serial chains
        be_.launch(nitems, SymKeys{ictx_, nitems, entA_});
        const uint64_t* sk = be_.sort_u64(entA_, entB_, nitems, 33);
        be_.launch(nitems, SymGather{sk, isym_, iunl_, nitems, gsym_});
        be_.launch(513, SymRunStart{sk, nitems, rstart_});
        be_.symrank(srstate_, gsym_, grank_, rstart_);
        be_.launch(nitems, SymScatter{sk, grank_, nitems, irank_});
        // static Huffman per chunk
        be_.memset(hw_, 0, (size_t)nchunks * kHwStride * 4);
        be_.launch(nitems, Hist{irank_, ial_, ienc_, nitems, hw_});
        be_.launch((size_t)nchunks * 3, HuffBuild{hw_, nchunks, hl_, hc_, hscr_});
        be_.launch(nitems, ItemBits{irank_, ial_, ienc_, irob_, hl_, nitems, blen_});
        be_.exclusive_scan_u32(blen_, bscan_, nitems);
        // bit packing
        std::vector<uint64_t> off(nchunks);
        for (uint32_t i = 0; i < nchunks; i++) off[i] = (uint64_t)i * kChunkCapWords;
        be_.h2d(outoff_, off.data(), nchunks * 8);
        be_.memset(out_, 0, (size_t)nchunks * kChunkCapWords * 4);
        be_.launch(nchunks, ChunkHeader{hl_, nchunks, nitems, len, ipos_, order_, ncounted_, stream_start_ ? 1 : 0, out_,
                                        outoff_, hdrbits_});
        be_.launch(nitems, Pack{irank_, ial_, ienc_, irob_, hl_, hc_, bscan_, hdrbits_, outoff_, nitems, out_});
        be_.launch(nchunks, ChunkTotals{bscan_, blen_, hdrbits_, nitems, nchunks, tot_});
        std::vector<uint32_t> tot(nchunks);
        be_.d2h(tot.data(), tot_, nchunks * 4);
        for (uint32_t i = 0; i < nchunks; i++) {
            size_t t = ((size_t)tot[i] + 31) / 32 * 4;  // finish pads to 32 bits, src/coder.rs:75-82
            if (t / 4 > kChunkCapWords) throw std::runtime_error("chunk payload overflow");
            size_t v = t;  // write_len, src/ioutil.rs:79-88
            while (v >= 128) { out.push_back((uint8_t)(128 + v % 128)); v /= 128; }
            out.push_back((uint8_t)v);
            size_t at = out.size();
            out.resize(at + t);
            be_.d2h(out.data() + at, out_ + off[i], t);
        }
        // ---- model state carried to the next block
        be_.d2d(ctxcount_, base_ + (size_t)nseg * 256, 256 * 4);
        uint8_t ltf;
        be_.d2h(&ltf, lt_[c] + nseg, 1);
        be_.memset(wlast_, 0, 32768 * 4);
        be_.launch((size_t)n + 1, WordsLast{win, E_[c], len, wlast_});
        be_.launch(32768, WordsApply{win, wlast_, len, (uint32_t)ltf, wsnap_});
        lt_carry_ = ltf;
        stream_start_ = false;
        be_.sync();
        stats.t_post += be_.now() - t2;
        stats.blocks++;
        stats.items += nitems;
        stats.chunks += nchunks;
        stats.in_bytes += n;
    }

    // window slide + LZEncoder::forward (src/lib.rs:83-84, src/lz.rs:82-87, src/matcher.rs:82-87):
    // the last kPre bytes move to offset 0, every position is rebased by 2^24, position 0 dies.
    // `slide_window` false = the caller re-uploads the whole window itself (object-level API).
    void slide(bool slide_window = true) {
        if (slide_window) be_.d2d(dwin(), dwin() + kNewMax, kPre);
        const int c = cur_;
        be_.launch(kPre, SlideArray<uint8_t>{S_[c], S_[c]});
        be_.launch(kPre, SlideArray<uint8_t>{ML_[c], ML_[c]});
        be_.launch(kPre, SlideArray<uint32_t>{ORD_[c], ORD_[c]});
        be_.launch(kPre, SlideArray<uint8_t>{LENMIN_, LENMIN_});
        be_.d2d(S_[c ^ 1], S_[c], kPre);
        be_.d2d(ML_[c ^ 1], ML_[c], kPre);
        be_.d2d(ORD_[c ^ 1], ORD_[c], (size_t)kPre * 4);
        be_.sync();
    }

    EncodeStats stats;
    static constexpr uint32_t kMaxChunks = 17;

   private:
    BE& be_;
    Cfg cfg_;
    uint32_t seg_, wsegs_, nseg_max_;
    int cur_ = 0;
    uint8_t lt_carry_ = kTyLit;
    bool stream_start_ = true;
    uint8_t* winbuf_;
    uint8_t *S_[2], *E_[2], *ML_[2];
    uint32_t* ORD_[2];
    uint32_t* first_[2];
    uint8_t* lt_[2];
    uint16_t* LR_;
    uint32_t* SRC_;
    uint8_t *W0_, *TY_, *LENMIN_, *LMV_;
    uint32_t *idx_, *kidx_;
    uint64_t *entA_, *entB_, *kentA_, *kentB_;
    uint32_t *f32_, *sc32_, *hpos_, *base_, *hist_, *csum_;
    uint16_t* lcnt_;
    uint32_t *ctxcount_, *fchg_;
    uint8_t* wsnap_;
    uint32_t* wlast_;
    uint32_t* ipos_;
    uint16_t *isym_, *ictx_, *irank_, *irob_, *grank_;
    uint8_t *iunl_, *ienc_, *ial_;
    uint32_t *gsym_, *blen_, *bscan_, *rstart_, *counts_;
    uint16_t* order_;
    uint32_t* ncounted_;
    uint16_t* srstate_;
    uint32_t* hw_;
    uint8_t* hl_;
    uint16_t* hc_;
    uint32_t *hscr_, *hdrbits_, *tot_;
    uint64_t* outoff_;
    uint32_t* out_;
};

// orz::encode (src/lib.rs:58-92) over a memory buffer that the backend can read with h2d():
// fills the window block by block, frames chunks, slides, and appends the EOF chunk.
template <class BE>
void encode_stream(StreamEncoder<BE>& enc, BE& be, const uint8_t* src, size_t n, bool src_on_device,
                   std::vector<uint8_t>& out) {
    enc.reset();
    size_t off = 0;
    while (off < n) {
        uint32_t take = (uint32_t)std::min<size_t>(n - off, kNewMax);
        if (src_on_device) be.d2d(enc.dwin() + kPre, src + off, take);
        else be.h2d(enc.dwin() + kPre, src + off, take);
        enc.encode_block(take, out);
        off += take;
        if (off < n) enc.slide();
    }
    out.push_back(0);  // EOF chunk, src/lib.rs:89
    enc.stats.out_bytes = out.size();
}

}  // namespace orz
