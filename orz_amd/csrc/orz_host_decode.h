// orz_host_decode.h -- host-side orz decoder of the product library.
//
// Decoding an orz stream is one serial chain per stream (every symbol's context depends on the
// bytes decoded before it), so it stays on the host (SURVEY.md 3.2, 8f row 3 lists a GPU decoder as
// a later row).  This is the product's own implementation of the reference's LZDecoder::decode /
// orz::decode (/root/reference/src/lz.rs:352-479, src/lib.rs:94-129, src/coder.rs:91-217,
// src/huffman.rs:118-167, src/symrank.rs:49-97, src/matcher.rs:62-91); it shares no code with the
// test oracle.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "orz_common.h"

namespace orz {
namespace host {

struct InvalidData : std::runtime_error {
    InvalidData() : std::runtime_error("invalid orz data") {}
};

// MSB-first bit reader over big-endian 32-bit words (src/coder.rs:159-217)
class BitReader {
   public:
    BitReader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    uint32_t bits(unsigned k) {  // k <= 32
        if (k == 0) return 0;
        fill();
        uint32_t v = (uint32_t)(acc_ >> (have_ - k)) & (k == 32 ? 0xffffffffu : ((1u << k) - 1));
        have_ -= k;
        return v;
    }
    uint32_t peek(unsigned k) {
        if (k == 0) return 0;
        fill();
        return (uint32_t)(acc_ >> (have_ - k)) & ((1u << k) - 1);
    }
    void skip(unsigned k) { have_ -= k; }
    uint32_t varint() {  // src/coder.rs:101-113: 2 bits per payload bit, LSB first
        uint32_t v = 0;
        for (unsigned sh = 0;; sh++) {
            uint32_t b = bits(2);
            if (sh < 32) v |= (b & 1u) << sh;
            if (b < 2) return v;
            if (sh > 40) throw InvalidData();
        }
    }

   private:
    void fill() {
        if (have_ >= 32) return;
        uint32_t w = 0;  // reading past the chunk yields zeros: the encoder padded to 32 bits
        for (int i = 0; i < 4; i++) w = (w << 8) | (at_ + i < n_ ? p_[at_ + i] : 0);
        at_ += 4;
        acc_ = (acc_ << 32) | w;
        have_ += 32;
    }
    const uint8_t* p_;
    size_t n_, at_ = 0;
    uint64_t acc_ = 0;
    unsigned have_ = 0;
};

// canonical Huffman decoding table (src/coder.rs:124-141, src/huffman.rs:118-167)
struct HuffDecoder {
    std::vector<uint32_t> lut;  // sym << 8 | len, indexed by max_len peeked bits
    unsigned max_len = 0;
    void read(BitReader& br) {
        max_len = br.varint();
        if (max_len > 16) throw InvalidData();
        std::vector<uint8_t> lens;
        for (;;) {
            uint32_t d = br.varint();
            if (d == 0) break;
            if (lens.size() + d > 4096) throw InvalidData();
            lens.resize(lens.size() + d - 1, 0);
            uint32_t sub = br.varint();
            if (sub > max_len) throw InvalidData();
            lens.push_back((uint8_t)(max_len - sub));
        }
        lut.assign((size_t)1 << max_len, 0);
        uint32_t code = 0;
        unsigned cur = 1;
        for (unsigned L = 1; L <= max_len; L++)
            for (size_t sy = 0; sy < lens.size(); sy++) {
                if (lens[sy] != L) continue;
                if (L > cur) { code <<= (L - cur); cur = L; }
                const unsigned rest = max_len - L;
                const size_t base = (size_t)code << rest;
                if (base + ((size_t)1 << rest) > lut.size()) throw InvalidData();
                for (size_t i = 0; i < ((size_t)1 << rest); i++) lut[base + i] = (uint32_t)(sy << 8) | L;
                code++;
            }
    }
    uint32_t sym(BitReader& br) const {
        if (max_len == 0) return 0;
        uint32_t e = lut[br.peek(max_len)];
        br.skip(e & 0xff);
        return e >> 8;
    }
};

class Decoder {  // LZDecoder, src/lz.rs:348-479
   public:
    Decoder() : ring_pos_(256 * kRing, 0), ring_min_(256 * kRing, 0), ring_exp_(256 * kRing, 0), head_(256, 0),
                rank_val_(512 * kSyms), rank_idx_(512 * kSyms), rank_cnt_(512, 0), rank_sum_(512, 1000000),
                words_(32768 * 2, 0) {}

    // back to the state of a new LZDecoder (src/lz.rs:352-357) without giving the arrays back
    void reset() {
        std::fill(ring_pos_.begin(), ring_pos_.end(), 0u);
        std::fill(ring_min_.begin(), ring_min_.end(), (uint8_t)0);
        std::fill(ring_exp_.begin(), ring_exp_.end(), (uint8_t)0);
        std::fill(head_.begin(), head_.end(), 0u);
        std::fill(rank_cnt_.begin(), rank_cnt_.end(), 0u);
        std::fill(rank_sum_.begin(), rank_sum_.end(), 1000000u);
        std::fill(words_.begin(), words_.end(), (uint8_t)0);
        first_ = true;
        after_literal_ = true;
    }

    // Bucket::forward for all contexts, src/lz.rs:359-364, src/matcher.rs:82-87
    void forward(size_t forward_len) {
        for (auto& p : ring_pos_) p = p > forward_len ? (uint32_t)(p - forward_len) : 0;
    }

    // decodes one chunk into sbuf at spos; returns min(spos_end, end_spos field)
    size_t decode(const uint8_t* tbuf, size_t tlen, uint8_t* sbuf, size_t spos) {
        BitReader br(tbuf, tlen);
        if (first_) {  // src/lz.rs:372-392
            uint32_t k = br.varint();
            if (k > kSyms) throw InvalidData();
            std::vector<uint16_t> vs;
            std::vector<bool> seen(kSyms, false);
            for (uint32_t i = 0; i < k; i++) {
                uint32_t v = br.bits(9);
                if (v >= kSyms || seen[v]) throw InvalidData();
                seen[v] = true;
                vs.push_back((uint16_t)v);
            }
            for (uint32_t i = 0; i < kSyms; i++)
                if (!seen[i]) vs.push_back((uint16_t)i);
            for (uint32_t c = 0; c < 512; c++)
                for (uint32_t i = 0; i < kSyms; i++) {
                    rank_val_[c * kSyms + i] = vs[i];
                    rank_idx_[c * kSyms + vs[i]] = (uint16_t)i;
                }
            first_ = false;
        }
        const size_t end_field = br.varint();
        const size_t n_items = br.varint();
        HuffDecoder t0, t1, t2;
        t0.read(br);
        t1.read(br);
        t2.read(br);
        for (size_t it = 0; it < n_items; it++) {
            if (spos + kMaxLen + 4 > (size_t)kBlock + kSent) throw InvalidData();
            const uint32_t r = (after_literal_ ? t1 : t0).sym(br);
            if (r >= kSyms) throw InvalidData();
            const uint32_t ctx = hash1(sbuf, (uint32_t)spos - 1);
            const uint32_t wkey = hash2(sbuf, (uint32_t)spos - 1);
            const uint8_t w0 = words_[wkey * 2], w1 = words_[wkey * 2 + 1];
            const uint32_t sym = unrank(ctx | (after_literal_ ? 256u : 0u), r, w0);
            if (sym == kWordSym) {
                ring_insert(ctx, spos, 0, 0);
                after_literal_ = false;
                sbuf[spos] = w0;
                sbuf[spos + 1] = w1;
                spos += 2;
            } else if (sym < 256) {
                ring_insert(ctx, spos, 0, 0);
                after_literal_ = true;
                sbuf[spos++] = (uint8_t)sym;
                set_word(sbuf, spos);
            } else {
                const uint32_t roid = (sym - 256) / 6, lenid = (sym - 256) % 6;
                uint32_t base = 0;
                for (uint32_t i = 0; i < roid; i++) base += 1u << (i >> 1);
                const uint32_t ro = base + br.bits(roid >> 1);
                if (ro >= kRing) throw InvalidData();
                const uint32_t node = (head_[ctx] + kRing - ro) % kRing;
                const uint32_t enc = lenid == 5 ? t2.sym(br) : lenid;
                const size_t src = ring_pos_[ctx * kRing + node];
                const uint32_t mn = std::max<uint32_t>(ring_min_[ctx * kRing + node], kMinLen);
                const uint32_t ex = std::max<uint32_t>(ring_exp_[ctx * kRing + node], kMinLen);
                const uint32_t len = enc + mn > ex ? enc + mn : (enc > 0 ? enc + mn - 1 : ex);  // src/lz.rs:459-467
                if (src >= spos || len > kMaxLen + 127) throw InvalidData();
                ring_insert(ctx, spos, ro, len);
                after_literal_ = false;
                for (uint32_t i = 0; i < len; i++) sbuf[spos + i] = sbuf[src + i];  // overlap-safe forward copy
                spos += len;
                set_word(sbuf, spos);
            }
        }
        return spos < end_field ? spos : end_field;
    }

   private:
    void set_word(const uint8_t* sbuf, size_t spos) {  // words[hash2(spos-3)] = sbuf[spos-2..spos]
        const uint32_t k = hash2(sbuf, (uint32_t)spos - 3);
        words_[k * 2] = sbuf[spos - 2];
        words_[k * 2 + 1] = sbuf[spos - 1];
    }
    void ring_insert(uint32_t ctx, size_t pos, uint32_t ro, uint32_t len) {  // Bucket::update, src/matcher.rs:62-80
        const uint32_t h = head_[ctx], nh = (h + 1) % kRing;
        if (len >= kMinLen) {
            const uint32_t ni = (h + kRing - ro) % kRing;
            uint8_t& m = ring_min_[ctx * kRing + ni];
            if (m <= len) m = (uint8_t)std::min<uint32_t>(len + 1, 127);
        }
        ring_pos_[ctx * kRing + nh] = (uint32_t)pos;
        ring_min_[ctx * kRing + nh] = 0;
        ring_exp_[ctx * kRing + nh] = (uint8_t)len;
        head_[ctx] = nh;
    }
    // SymRankCoder::decode + update, src/symrank.rs:49-97
    uint32_t unrank(uint32_t c, uint32_t r, uint32_t unlikely) {
        uint16_t* val = &rank_val_[c * kSyms];
        uint16_t* idx = &rank_idx_[c * kSyms];
        const uint32_t iu = idx[unlikely];
        uint32_t i = r == kSyms - 1 ? iu : r + (r >= iu ? 1u : 0u);
        if (i >= kSyms) throw InvalidData();
        const uint16_t v = val[i];
        uint32_t& cnt = rank_cnt_[c];
        uint32_t& sum = rank_sum_[c];
        if (cnt > kSyms) { cnt = cnt * 9 / 10; sum = sum * 9 / 10; }
        cnt += 1;
        sum += i;
        const uint32_t dec = (i / 16 + (uint16_t)(sum / 16 / cnt)) & 0xffff;
        uint32_t ni = i > dec ? i - dec : 0;
        if (ni < i / 2) ni = i / 2;
        const uint32_t n = i - ni;
        if (n == 1) {
            const uint16_t o = val[ni];
            val[i] = o; idx[o] = (uint16_t)i;
            val[ni] = v; idx[v] = (uint16_t)ni;
        } else if (n > 1) {
            const uint32_t mid = ni + n / 2;
            const uint16_t a = val[mid], b2 = val[ni];
            val[i] = a; idx[a] = (uint16_t)i;
            val[mid] = b2; idx[b2] = (uint16_t)mid;
            val[ni] = v; idx[v] = (uint16_t)ni;
        }
        return v;
    }

    std::vector<uint32_t> ring_pos_;
    std::vector<uint8_t> ring_min_, ring_exp_;
    std::vector<uint32_t> head_;
    std::vector<uint16_t> rank_val_, rank_idx_;
    std::vector<uint32_t> rank_cnt_, rank_sum_;
    std::vector<uint8_t> words_;
    bool first_ = true, after_literal_ = true;
};

// the buffers of one orz::decode call (src/lib.rs:99-101), reusable across the members of a container
struct DecodeWorkspace {
    Decoder dec;
    std::vector<uint8_t> win, tbuf;
    bool used = false, slid = false;
    DecodeWorkspace() : win((size_t)kBlock * 2 + 2 * kSent, 0), tbuf((size_t)kPre * 3) {}
    void begin_stream() {
        if (!used) { used = true; return; }
        dec.reset();
        // bytes at and after SBVEC_PREMATCH_LEN are always written before they are read; the history below it
        // is zero unless the previous stream slid its window there
        if (slid) std::fill(win.begin(), win.begin() + kSent + kPre, (uint8_t)0);
        slid = false;
    }
};

// orz::decode over callbacks (src/lib.rs:94-129).  read(buf, n) must fill exactly n bytes or return false.
template <class ReadExact, class WriteAll, class Progress>
void decode_stream(DecodeWorkspace& ws, ReadExact&& rd, WriteAll&& wr, Progress&& progress) {
    ws.begin_stream();
    Decoder& dec = ws.dec;
    std::vector<uint8_t>& tbuf = ws.tbuf;
    uint8_t* sbuf = ws.win.data() + kSent;
    size_t spos = kPre, in_total = 0, out_total = 0;
    for (;;) {
        size_t t = 0;
        for (unsigned sh = 0;; sh += 7) {  // read_len, src/ioutil.rs:60-77
            uint8_t b;
            if (!rd(&b, 1)) throw InvalidData();
            in_total++;
            t |= (size_t)(b & 0x7f) << sh;
            if (!(b & 0x80)) break;
            if (sh > 56) throw InvalidData();
        }
        if (t == 0) break;
        if (t >= tbuf.size()) throw InvalidData();  // src/lib.rs:111-113
        if (!rd(tbuf.data(), t)) throw InvalidData();
        in_total += t;
        const size_t end = dec.decode(tbuf.data(), t, sbuf, spos);
        if (end < spos) throw InvalidData();
        wr(sbuf + spos, end - spos);
        out_total += end - spos;
        spos = end;
        if (spos >= kBlock) {  // src/lib.rs:120-125
            std::memmove(sbuf, sbuf + (kBlock - kPre), kPre);
            dec.forward(kBlock - kPre);
            ws.slid = true;
            progress(false, in_total, out_total);
            spos = kPre;
        }
    }
    progress(true, in_total, out_total);
}
template <class ReadExact, class WriteAll, class Progress>
void decode_stream(ReadExact&& rd, WriteAll&& wr, Progress&& progress) {
    DecodeWorkspace ws;
    decode_stream(ws, rd, wr, progress);
}

}  // namespace host
}  // namespace orz
