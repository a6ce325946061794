// emu_backend.cpp -- host emulation of the device backend (TEST INFRASTRUCTURE ONLY).
// Runs the very same kernel bodies (orz_amd/csrc/orz_kernels.h, orz_parse.h) and orchestration
// (orz_stream.h) on the CPU, so the CPU-only test tier can check the encoder's logic byte for
// byte against the oracle without a GPU.  Thread kernels run as host loops; the wave-cooperative
// parse kernel runs on the SIMT emulator of simt.h, with its blocks visited in ascending,
// descending (= pure Jacobi speculation) or shuffled order.  Never linked into the product.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>

#include "../../orz_amd/csrc/orz_decode_check.h"
#include "../../orz_amd/csrc/orz_decode_device.h"
#include "../../orz_amd/csrc/orz_stream.h"
#include "simt.h"
#if defined(ORZ_EMU_THREADS)  // the race check (tests/race): a launch's threads / blocks on this many host threads
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#endif

namespace {
struct EmuBackend {
    int order = 1;  // simt::Order for wave kernels
    uint64_t seed = 1;
    // (allocations the encoder asks for without a zero fill come back POISONED: the device hands out recycled memory, and
    // whatever the kernels read before they have written it must not matter)
    // ORZ_EMU_ARENA_MB=<n>: every buffer carved out of one block (2 MiB-aligned when large, 256 B otherwise, like the HIP
    // backend's ORZ_ARENA_MB), the block filled with pseudo-random bytes first: neighbours instead of unmapped slack behind
    // every buffer, plausible garbage instead of a constant in what was not asked to be zeroed -- out-of-bounds and
    // uninitialised reads change the output instead of passing unnoticed.
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    void* arena_take(size_t bytes) {
        if (!arena) {
            const char* v = std::getenv("ORZ_EMU_ARENA_MB");
            if (!v || !std::atoi(v)) return nullptr;
            arena_bytes = (size_t)std::atoi(v) << 20;
            arena = (char*)std::malloc(arena_bytes);
            if (const char* f = std::getenv("ORZ_EMU_ARENA_FILL")) {  // (a constant instead: which of the two effects is it?)
                std::memset(arena, std::atoi(f), arena_bytes);
            } else {
                uint64_t x = 0x9E3779B97F4A7C15ull;
                for (size_t i = 0; i + 8 <= arena_bytes; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::memcpy(arena + i, &x, 8); }
            }
        }
        const size_t align = bytes >= (1u << 20) ? (2u << 20) : 256;
        const size_t at = (arena_used + align - 1) / align * align;
        if (at + bytes > arena_bytes) return nullptr;
        arena_used = at + bytes;
        return arena + at;
    }
    ~EmuBackend() { std::free(arena); }
    long fail_alloc_in = -1;  // (tests) the allocation that many calls from now throws std::bad_alloc; -1 = never
    template <class T> T* alloc(size_t n, bool zero = true) {
        if (fail_alloc_in == 0) { fail_alloc_in = -1; throw std::bad_alloc(); }
        if (fail_alloc_in > 0) fail_alloc_in--;
        if (void* q = arena_take((n ? n : 1) * sizeof(T))) {
            if (zero) std::memset(q, 0, (n ? n : 1) * sizeof(T));
            return (T*)q;
        }
        if (zero) return (T*)std::calloc(n ? n : 1, sizeof(T));
        T* p = (T*)std::malloc((n ? n : 1) * sizeof(T));
        if (p) std::memset(p, std::getenv("ORZ_EMU_POISON") ? std::atoi(std::getenv("ORZ_EMU_POISON")) : 0xA5, (n ? n : 1) * sizeof(T));
        return p;
    }
    void free(void* p) {
        if (arena && (char*)p >= arena && (char*)p < arena + arena_bytes) return;
        std::free(p);
    }
    void release_arena() {}
    void parse_token_acquire() {}
    void parse_token_release() {}
    void memset(void* p, int v, size_t n) { std::memset(p, v, n); }
    void poison(void* p, size_t n) { std::memset(p, 0xA5, n); }
    void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    void h2d_pinned(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    void d2h(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    void d2h_async(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    uint64_t take_host_syncs() { return 0; }
    void d2d(void* d, const void* s, size_t n) { std::memmove(d, s, n); }
    void sync() {}
    void mark_end() {}
    bool check_hints() const { return true; }  // (the emulation verifies every count the host derives)
    uint32_t handoff_polls() const { return 1; }  // blocks run one after another here: waiting cannot help
    uint32_t handoff_deadline() const { return 0; }
    uint32_t near_blocks() const { return 16; }  // small windows here: exercise the far-wave paths too
    uint32_t far_deadline() const { return 0; }
    uint32_t skip_after() const { return 0; }
    void select(int) {}
    void record(int) {}
    void wait(int) {}
    bool graphs_enabled() const { return false; }
    bool graph_replay(uint64_t) { return false; }
    void graph_capture_begin() {}
    void graph_capture_end(uint64_t) {}
    void graph_capture_abort() {}
    void timed_begin(int = 0) {}
    void timed_end(int = 0) {}
    void set_timing(bool) {}
    double collect_timed(uint64_t* n, double* ms = nullptr, uint64_t* cnt = nullptr) {
        if (n) *n = 0;
        for (int i = 0; i < 4; i++) { if (ms) ms[i] = 0; if (cnt) cnt[i] = 0; }
        return 0.0;
    }
    double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#if defined(ORZ_EMU_THREADS)
    // Race check: wavefront w of a launch (threads 64 w .. 64 w + 63, one after another) runs on host thread w % T, so that any
    // two threads of different wavefronts are unordered to ThreadSanitizer exactly as they are on the GPU.  A kernel whose
    // threads read what other threads of the SAME launch write -- other than through the atomics -- is reported, whatever
    // the timing of this run was.
    // (a pool, not a thread per launch: a block has thousands of launches and the sanitizer's thread table is finite; the
    // hand-over through the mutex orders the launches, as a HIP stream does)
    struct Pool {
        std::mutex m;
        std::condition_variable cv, done_cv;
        std::function<void(unsigned)> job;
        uint64_t gen = 0;
        unsigned pending = 0;
        bool stop = false;
        std::vector<std::thread> th;
        Pool() {
            for (unsigned t = 1; t < ORZ_EMU_THREADS; t++)
                th.emplace_back([this, t] {
                    uint64_t seen = 0;
                    for (;;) {
                        std::function<void(unsigned)> j;
                        {
                            std::unique_lock<std::mutex> lk(m);
                            cv.wait(lk, [&] { return stop || gen != seen; });
                            if (stop) return;
                            seen = gen;
                            j = job;
                        }
                        j(t);
                        std::lock_guard<std::mutex> lk(m);
                        if (--pending == 0) done_cv.notify_all();
                    }
                });
        }
        ~Pool() {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            cv.notify_all();
            for (auto& x : th) x.join();
        }
        void run(const std::function<void(unsigned)>& body) {
            { std::lock_guard<std::mutex> lk(m); job = body; pending = ORZ_EMU_THREADS - 1; gen++; }
            cv.notify_all();
            body(0u);
            std::unique_lock<std::mutex> lk(m);
            done_cv.wait(lk, [&] { return pending == 0; });
        }
    };
    template <class Body> static void on_threads(const Body& body) {
        static Pool pool;
        pool.run(body);
    }
    template <class F> void launch(size_t n, const F& f) {
        const size_t nw = (n + 63) / 64;
        if (nw < 2) { for (size_t i = 0; i < n; i++) f(i); return; }
        on_threads([&](unsigned t) {
            for (size_t w = t; w < nw; w += ORZ_EMU_THREADS)
                for (size_t i = w * 64; i < n && i < (w + 1) * 64; i++) f(i);
        });
    }
    template <class K> void launch_waves(size_t nblocks, const K& k, size_t lds_bytes) {
        const uint64_t sd = seed++;
        if (nblocks < 2) { simt::launch_waves(nblocks, k, lds_bytes, (simt::Order)order, sd); return; }
        on_threads([&](unsigned t) { simt::launch_waves(nblocks, k, lds_bytes, (simt::Order)order, sd, t, ORZ_EMU_THREADS); });
    }
    template <class K> void launch_group(const K& k) {
        const size_t lds = k.lds_bytes();
        std::vector<uint8_t> mem(lds + 64);
        for (uint32_t ph = 0; ph < K::kPhases; ph++)
            on_threads([&](unsigned t) { for (uint32_t x = t; x < 1024; x += ORZ_EMU_THREADS) k.phase(ph, x, 1024, mem.data(), lds != 0); });
    }
#else
    template <class F> void launch(size_t n, const F& f) {
        for (size_t i = 0; i < n; i++) f(i);
    }
    template <class K> void launch_waves(size_t nblocks, const K& k, size_t lds_bytes) {
        simt::launch_waves(nblocks, k, lds_bytes, (simt::Order)order, seed++);
    }
    template <class K> void launch_group(const K& k) {
        const size_t lds = k.lds_bytes();
        std::vector<uint8_t> mem(lds + 64);
        for (uint32_t ph = 0; ph < K::kPhases; ph++)
            for (uint32_t t = 0; t < 1024; t++) k.phase(ph, t, 1024, mem.data(), lds != 0);
    }
#endif
    void huffbuild(const orz::HuffBuild& f) { if (f.nchunks) launch_waves((size_t)f.nchunks * 3, orz::HuffWave{f}, orz::HuffWave::lds_bytes()); }
    void rank(const orz::RankArgs& a, uint32_t nchunks) {
        // one block per chunk, 256 threads around one barrier: run each block as two thread loops
        std::vector<uint32_t> rows((orz::kRankChunk + 1) * 256);
        for (uint32_t ch = nchunks; ch-- > 0;) {  // chunk 0 moves the front: run it last
            // phase split at the barrier: emulate by running every thread up to the barrier first
            // threads are independent before the barrier (own column) and read rows[] after it
            for (int phase = 0; phase < 2; phase++)
                for (uint32_t c = 256; c-- > 0;) run_rank_thread(a, ch, c, rows.data(), phase);
        }
        // the remaining blocks of the launch refresh the bitmap summaries
        struct Stop {};
        for (int which = 0; which < 2; which++) {
            const uint64_t* L0 = which ? a.kbits : a.vbits;
            uint64_t* L1 = which ? a.k1 : a.v1;
            uint64_t* L2 = which ? a.k2 : a.v2;
            const uint32_t nw = which ? a.nkwords : a.nvwords;
            for (uint32_t blk = 0; blk < (nw + 4095) / 4096; blk++)
                for (int phase = 0; phase < 2; phase++)
                    for (uint32_t t = 0; t < 256; t++) {
                        if (phase == 0) {
                            try { orz::rebuild_summaries(L0, L1, L2, nw, blk, t, (uint64_t*)rows.data(), []() { throw Stop(); }); } catch (Stop&) {}
                        } else {
                            orz::rebuild_summaries(L0, L1, L2, nw, blk, t, (uint64_t*)rows.data(), []() {});
                        }
                    }
        }
    }
    // executes thread c of chunk ch either up to the barrier (phase 0) or from it (phase 1)
    static void run_rank_thread(const orz::RankArgs& a, uint32_t ch, uint32_t c, uint32_t* rows, int phase) {
        struct Stop {};
        int seen = 0;
        auto sync = [&]() { seen++; if (phase == 0) throw Stop(); };
        if (phase == 0) {
            try { orz::rank_chunk(a, ch, c, rows, sync); } catch (Stop&) {}
        } else {
            // re-run from the start with side effects of the pre-barrier part being idempotent
            orz::rank_chunk(a, ch, c, rows, [&]() {});
        }
    }
    const uint64_t* sort_u64(uint64_t* a, uint64_t*, size_t n, int bits, int begin_bit = 0) {
        uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
        std::stable_sort(a, a + n, [mask, begin_bit](uint64_t x, uint64_t y) { return ((x & mask) >> begin_bit) < ((y & mask) >> begin_bit); });
        return a;
    }
    void sort_by_ctx(const uint16_t* ctx, uint16_t* ctx_sorted, uint32_t* perm, size_t n) {
        std::iota(perm, perm + n, 0u);
        std::stable_sort(perm, perm + n, [&](uint32_t x, uint32_t y) { return ctx[x] < ctx[y]; });
        for (size_t i = 0; i < n; i++) ctx_sorted[i] = ctx[perm[i]];
    }
    void sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits) {
        uint32_t mask = bits >= 32 ? ~0u : ((1u << bits) - 1);
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return (kin[x] & mask) < (kin[y] & mask); });
        for (size_t i = 0; i < n; i++) { kout[i] = kin[perm[i]]; vout[i] = vin[perm[i]]; }
    }
    void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        uint32_t run = 0;
        for (size_t i = 0; i < n; i++) { uint32_t v = in[i]; out[i] = run; run += v; }
    }
    void inclusive_max_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        uint32_t run = 0;
        for (size_t i = 0; i < n; i++) { if (in[i] > run) run = in[i]; out[i] = run; }
    }
    void symrank(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank, const uint32_t* rstart, uint32_t nitems, uint32_t* flags, uint16_t*, uint16_t*) {
        flags[0] = flags[1] = flags[2] = 0;  // (the reference loop below needs no second run; the check runs all the same)
        for (uint32_t c = 0; c < 512; c++) {
            uint16_t value[orz::kSyms], index[orz::kSyms];
            orz::symrank_run(value, index, srstate + (size_t)c * orz::kSrWords, gsym, grank, rstart[c], rstart[c + 1]);
        }
        launch(orz::SymCheck::kThreads, orz::SymCheck{gsym, grank, nitems, flags + 1, nullptr});
    }
};
}  // namespace

// order: 0 ascending, 1 descending, 2 shuffled block order inside a sweep
extern "C" int emu_encode(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, unsigned seg, unsigned win,
                          int order, uint8_t** dst, size_t* dst_len, unsigned long long* stats5) {
    try {
        EmuBackend be;
        be.order = order;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, seg, win ? win : 4096);
        std::vector<uint8_t> out;
        orz::encode_stream(enc, be, src, n, false, out);
        *dst = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        if (stats5) {
            stats5[0] = enc.stats.blocks; stats5[1] = enc.stats.sweeps; stats5[2] = enc.stats.seg_evals;
            stats5[3] = enc.stats.items; stats5[4] = enc.stats.chunks;
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_encode: %s\n", e.what());
        return -1;
    }
}
static std::string g_emu_err;
extern "C" const char* emu_last_error() { return g_emu_err.c_str(); }  // message of the last emu_encode_fast that returned -1
// the fast parse mode (orz_fast.h) on the emulation backend
extern "C" int emu_encode_fast(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, unsigned tile, unsigned rounds,
                               uint8_t** dst, size_t* dst_len, unsigned long long* stats5) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, tile ? tile : orz::kFastTile, rounds ? rounds : orz::kFastRounds);
        std::vector<uint8_t> out;
        orz::encode_stream(enc, be, src, n, false, out);
        *dst = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        if (stats5) {
            stats5[0] = enc.stats.blocks; stats5[1] = enc.stats.sweeps; stats5[2] = enc.stats.seg_evals;
            stats5[3] = enc.stats.items; stats5[4] = enc.stats.chunks;
        }
        if (std::getenv("ORZ_EVAL_STATS"))
            std::fprintf(stderr, "eval: %llu positions visited, %llu evaluated (first round %llu, dirty %llu, far due %llu), %llu settled the ring end with positions; verify: %llu skipped would differ (%llu in lwm)\n", orz::g_eval_stats[0],
                         orz::g_eval_stats[1], orz::g_eval_stats[2], orz::g_eval_stats[3], orz::g_eval_stats[4], orz::g_eval_stats[5], orz::g_eval_stats[6], orz::g_eval_stats[7]);
        if (std::getenv("ORZ_SCAN_HIST")) {
            for (int r = 0; r < 4; r++) {
                std::fprintf(stderr, "scan hist %d:", r);
                for (int k = 0; k < 64; k++) std::fprintf(stderr, " %llu", orz::g_scan_hist[r][k]);
                std::fprintf(stderr, "\n");
            }
        }
        if (std::getenv("ORZ_FAR_STATS"))
            std::fprintf(stderr, "compact lists: %llu scans, %llu records read, %llu of them continued in the window; %llu trips below the window\n",
                         orz::g_far_stats[0], orz::g_far_stats[1], orz::g_far_stats[2], orz::g_far_stats[3]);
        return 0;
    } catch (const std::exception& e) {
        g_emu_err = e.what();
        if (!std::getenv("ORZ_VERIFY_INJECT")) std::fprintf(stderr, "emu_encode_fast: %s\n", e.what());
        return -1;
    }
}
// the same through the device-output path (round 6: FrameChunks / FrameAdvance / FrameEof frame the stream in "device" memory);
// cap = 0: the encoder's own buffer, else a caller buffer of that many bytes.  Returns -1 with the message in emu_last_error().
extern "C" int emu_encode_fast_device(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, size_t cap, uint8_t** dst, size_t* dst_len) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        std::vector<uint8_t> mine(cap);
        const auto r = orz::encode_stream_device(enc, be, src, n, false, cap ? mine.data() : nullptr, cap);
        *dst = (uint8_t*)std::malloc(r.len ? r.len : 1);
        std::memcpy(*dst, r.data, r.len);
        *dst_len = r.len;
        return 0;
    } catch (const std::exception& e) {
        g_emu_err = e.what();
        return -1;
    }
}
// An allocation fails in the middle of a tail set's growth (the `fail_in`-th allocation after the encoder was built), the encode
// fails -- and the SAME encoder then encodes `src` again: it must write what a fresh encoder writes (a half-grown
// set left null pointers behind an unchanged capacity).  Returns 0 and the second stream; -2 if the first encode did not fail.
extern "C" int emu_encode_fast_after_failed_growth(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, long fail_in,
                                                   uint8_t** dst, size_t* dst_len) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        std::vector<uint8_t> out;
        be.fail_alloc_in = fail_in;
        bool failed = false;
        try { orz::encode_stream(enc, be, src, n, false, out); } catch (const std::bad_alloc&) { failed = true; }
        be.fail_alloc_in = -1;
        if (!failed) return -2;
        out.clear();
        orz::encode_stream(enc, be, src, n, false, out);
        *dst = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_encode_fast_after_failed_growth: %s\n", e.what());
        return -1;
    }
}
// two inputs through ONE encoder, the second stream returned: a reused encoder must write what a fresh one writes
extern "C" int emu_encode_fast_reused(const uint8_t* first, size_t n_first, const uint8_t* src, size_t n, int depth, int lazy1, int lazy2,
                                      uint8_t** dst, size_t* dst_len) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        std::vector<uint8_t> out;
        orz::encode_stream(enc, be, first, n_first, false, out);
        out.clear();
        orz::encode_stream(enc, be, src, n, false, out);
        *dst = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_encode_fast_reused: %s\n", e.what());
        return -1;
    }
}
// the fast parse's items (block, window offset, source, symbol, length, flags) next to its stream: parity forensics
struct EmuItem { uint32_t block, pos, src; uint16_t sym, rank, ctx; uint8_t mlen, al, unl, enc; };
extern "C" long emu_encode_fast_trace(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, uint8_t** dst, size_t* dst_len, EmuItem* items,
                                      size_t cap) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        orz::ItemTrace tr;
        enc.trace = &tr;
        std::vector<uint8_t> out;
        orz::encode_stream(enc, be, src, n, false, out);
        *dst = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        const size_t k = tr.pos.size() < cap ? tr.pos.size() : cap;
        for (size_t i = 0; i < k; i++)
            items[i] = EmuItem{tr.block[i], tr.pos[i], tr.src[i], tr.sym[i], tr.rank[i], tr.ctx[i], tr.mlen[i], tr.al[i], tr.unl[i], tr.enc[i]};
        return (long)tr.pos.size();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_encode_fast_trace: %s\n", e.what());
        return -1;
    }
}
// the two bytes in front of the window after a stream was encoded (the context of the oldest history position looks there)
extern "C" int emu_window_front(const uint8_t* src, size_t n, uint8_t* front2) {
    try {
        EmuBackend be;
        orz::Cfg cfg{15, 9, 6};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        std::vector<uint8_t> out;
        orz::encode_stream(enc, be, src, n, false, out);
        front2[0] = enc.dwin()[-2];
        front2[1] = enc.dwin()[-1];
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_window_front: %s\n", e.what());
        return -1;
    }
}
extern "C" void emu_free(void* p) { std::free(p); }
// the Huffman table kernel alone (orz_kernels.h, HuffWave): nchunks x kHwStride weights in, lengths and codes out
extern "C" int emu_huff_build(const uint32_t* hw, unsigned nchunks, uint8_t* hl, uint16_t* hc) {
    EmuBackend be;
    be.huffbuild(orz::HuffBuild{hw, nchunks, hl, hc});
    return (int)orz::kHwStride;
}

// the hand-off word of the parse kernel (orz_parse.h, ExitPair): pack, then unpack into out[4] = entry, exit, settled, sweep
extern "C" unsigned long long emu_exitpair(unsigned sweep, int settled, unsigned entry, unsigned exit, unsigned* out) {
    const uint64_t e = orz::ExitPair::make(sweep, settled != 0, entry, exit);
    out[0] = orz::ExitPair::entry(e);
    out[1] = orz::ExitPair::exit(e);
    out[2] = orz::ExitPair::settled(e) ? 1u : 0u;
    out[3] = orz::ExitPair::sweep(e);
    return e;
}

// the device decoder's kernel body and host driver on the CPU: members container -> bytes
// returns 0, or 1 with a message in err (cap bytes)
extern "C" int emu_decode_members(const uint8_t* src, size_t n, unsigned slots, uint8_t** dst, size_t* dst_len, size_t* members,
                                  char* err, size_t cap) {
    try {
        EmuBackend be;
        std::vector<uint8_t> out;
        orz::DecodeStats st;
        orz::decode_members_device(be, src, n, out, st, slots ? slots : 4);
        uint8_t* p = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        std::memcpy(p, out.data(), out.size());
        *dst = p; *dst_len = out.size(); *members = (size_t)st.members;
        return 0;
    } catch (const std::exception& e) {
        if (err && cap) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
        return 1;
    }
}

// The object-level seam (orz_lz_encoder_encode / _forward, orz_capi.hip) on the emulation: the caller keeps the window, slides
// it with copy_within (src/lib.rs:83: the pad in front of the window keeps its zeros) and hands it over for every block;
// `keep_front` = the upload after a forward() starts at window offset 0 (what the library does) instead of at the pad.
extern "C" int emu_encode_fast_seam(const uint8_t* src, size_t n, int depth, int lazy1, int lazy2, int keep_front, uint8_t** dst, size_t* dst_len) {
    try {
        EmuBackend be;
        orz::Cfg cfg{depth, lazy1, lazy2};
        orz::StreamEncoder<EmuBackend> enc(be, cfg, 62, 64, true, orz::kFastTile, orz::kFastRounds);
        std::vector<uint8_t> window((size_t)orz::kBlock + 2 * orz::kSent, 0), out;
        uint8_t* sbuf = window.data() + orz::kSent;
        bool slid = false;
        for (size_t off = 0; off < n;) {
            const size_t take = std::min<size_t>(orz::kNewMax, n - off);
            std::memcpy(sbuf + orz::kPre, src + off, take);
            const size_t sbuf_len = orz::kPre + take;
            if (slid && keep_front) be.h2d(enc.dwin(), sbuf, sbuf_len + orz::kSent);
            else be.h2d(enc.dwinbuf(), sbuf - orz::kSent, sbuf_len + 2 * orz::kSent);
            std::vector<size_t> ends;
            enc.encode_block((uint32_t)take, out, &ends);
            off += take;
            std::memmove(sbuf, sbuf + (orz::kBlock - orz::kPre), orz::kPre);
            enc.slide(false);
            slid = true;
        }
        out.push_back(0);
        *dst = (uint8_t*)std::malloc(out.size());
        std::memcpy(*dst, out.data(), out.size());
        *dst_len = out.size();
        return 0;
    } catch (const std::exception& e) {
        g_emu_err = e.what();
        std::fprintf(stderr, "emu_encode_fast_seam: %s\n", e.what());
        return -1;
    }
}

// the engine of ORZ_VERIFY=decode (orz_decode_check.h) alone: `stream` fed in pieces of `piece` bytes against `input`;
// returns 0 when every chunk decodes to the input and the stream ends where the input does, else 1 with the message in err
extern "C" int emu_decode_check(const uint8_t* stream, size_t n, const uint8_t* input, size_t m, size_t piece, char* err, size_t cap) {
    try {
        orz::host::DecodeCheck chk;
        if (!piece) piece = n ? n : 1;
        size_t fed = 0;
        for (size_t at = 0; at < n; at += piece) {
            // the input arrives block by block, ahead of the bytes that encode it (as in orz_encode)
            while (fed < m && fed < (at / piece + 2) * (size_t)orz::kNewMax) {
                const size_t k = std::min<size_t>(orz::kNewMax, m - fed);
                chk.feed_input(input + fed, k);
                fed += k;
            }
            chk.feed_output(stream + at, std::min(piece, n - at));
        }
        if (fed < m) chk.feed_input(input + fed, m - fed);
        chk.finish();
        return 0;
    } catch (const std::exception& e) {
        if (err && cap) { std::strncpy(err, e.what(), cap - 1); err[cap - 1] = 0; }
        return 1;
    }
}
