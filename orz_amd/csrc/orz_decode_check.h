// orz_decode_check.h -- the engine of ORZ_VERIFY=decode (orz_capi.hip): an orz stream, fed as it is produced, against the
// bytes it was encoded from, through the library's own decoder (orz_host_decode.h; LZDecoder::decode,
// /root/reference/src/lz.rs:366-478, driven like orz::decode, src/lib.rs:94-129).  Incremental: input and output arrive in
// pieces of any size, a chunk is decoded as soon as it is complete, what has been checked is dropped.
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "orz_host_decode.h"

namespace orz {
namespace host {

class DecodeCheck {
   public:
    DecodeCheck() : ws_(new DecodeWorkspace) { ws_->begin_stream(); }
    void feed_input(const uint8_t* p, size_t n) { in_.insert(in_.end(), p, p + n); }
    // bytes of the stream in order; throws std::runtime_error on the first chunk that does not decode to the input
    void feed_output(const uint8_t* p, size_t n) {
        out_.insert(out_.end(), p, p + n);
        for (;;) {
            size_t t = 0, at = opos_;
            unsigned sh = 0;
            bool whole = false;
            while (at < out_.size()) {  // read_len, src/ioutil.rs:60-77
                const uint8_t b = out_[at++];
                t |= (size_t)(b & 0x7f) << sh;
                sh += 7;
                if (!(b & 0x80)) { whole = true; break; }
            }
            if (!whole) break;
            if (t == 0) { eof_ = true; opos_ = at; break; }
            if (at + t > out_.size()) break;  // the chunk is not complete yet
            if (t >= ws_->tbuf.size()) bad("a chunk longer than the decoder accepts");
            std::memcpy(ws_->tbuf.data(), out_.data() + at, t);
            uint8_t* sbuf = ws_->win.data() + kSent;
            size_t end;
            try {
                end = ws_->dec.decode(ws_->tbuf.data(), t, sbuf, spos_);
            } catch (const std::exception&) {
                bad("the decoder rejects a chunk");
            }
            if (end < spos_) bad("the decoder rejects a chunk");
            const size_t got = end - spos_;
            if (ipos_ + got > in_.size() || std::memcmp(sbuf + spos_, in_.data() + ipos_, got) != 0) bad("a chunk decodes to other bytes than were encoded");
            ipos_ += got;
            checked_ += got;
            spos_ = end;
            if (spos_ >= kBlock) {  // src/lib.rs:120-125
                std::memmove(sbuf, sbuf + (kBlock - kPre), kPre);
                ws_->dec.forward(kBlock - kPre);
                spos_ = kPre;
            }
            opos_ = at + t;
            // drop what has been checked (the buffers stay small on long streams)
            if (opos_ > (1u << 24)) { out_.erase(out_.begin(), out_.begin() + (ptrdiff_t)opos_); opos_ = 0; }
            if (ipos_ > (1u << 24)) { in_.erase(in_.begin(), in_.begin() + (ptrdiff_t)ipos_); ipos_ = 0; }
        }
    }
    void finish() {
        if (!eof_ || ipos_ != in_.size() || opos_ != out_.size()) bad("the stream ends before its input does (or carries bytes behind its end)");
    }
    size_t checked() const { return checked_; }

   private:
    [[noreturn]] void bad(const char* what) {
        throw std::runtime_error(std::string("ORZ_VERIFY=decode: ") + what + " (after " + std::to_string(checked_) + " verified bytes): the encode fails, nothing of this block is handed out (earlier blocks of a streaming call already were: the output is incomplete)");
    }
    std::unique_ptr<DecodeWorkspace> ws_;
    std::vector<uint8_t> in_, out_;
    size_t ipos_ = 0, opos_ = 0, spos_ = kPre, checked_ = 0;
    bool eof_ = false;
};

}  // namespace host
}  // namespace orz
