// orz_capi.hip -- C ABI of liborz_hip.so (declared in include/orz_hip.h).
//
// Host side of the MI355X encoder: mirrors the reference's LZEncoder / orz::encode call surface
// (/root/reference/src/lz.rs:69-95, src/lib.rs:58-92) on top of StreamEncoder<HipBackend>.
// There is no CPU fallback: if no HIP device is usable every constructor fails with ORZ_ENODEV.
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/orz_hip.h"
#include "backend_hip.h"
#include "orz_decode_device.h"
#include "orz_host_decode.h"
#include "orz_decode_check.h"
#include "orz_stream.h"

namespace {
// An encoder drives three HIP streams (parse; symbol ranking; Huffman + packing) and the members interface runs several
// encoders side by side.  The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues
// (4 by default): streams that share a queue run their kernels one after another, so another member's 50 ms
// symbol-ranking launch stalls a parse behind it.  Measured on MI355X, 8 encoders, 64 MiB members, -l1: 4 queues
// 344 MB/s, 8: 385, 16: 443, 32: 437.  The variable is read when the HIP runtime initialises, so it is set here, when
// the library is loaded, unless the host process has chosen a value itself.
struct HwQueueDefault {
    HwQueueDefault() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hw_queue_default;


thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

orz::Cfg to_cfg(const orz_lzcfg* c) {
    return orz::Cfg{(int)c->match_depth, (int)c->lazy_match_depth1, (int)c->lazy_match_depth2};
}
bool cfg_ok(const orz_lzcfg* c) {
    return c && c->match_depth >= 1 && c->match_depth <= 200 && c->lazy_match_depth1 <= 200 &&
           c->lazy_match_depth2 <= 200;
}

constexpr unsigned kDefaultSeg = 62;  // window: 0 = as many segments as parse waves fit the device at once

unsigned env_u(const char* name, unsigned dflt) {
    const char* v = std::getenv(name);
    return v && *v ? (unsigned)std::strtoul(v, nullptr, 10) : dflt;
}

using Enc = orz::StreamEncoder<orz::HipBackend>;

// parse mode of a new encoder: the GPU-native fast mode unless ORZ_MODE=exact asks for the reference-identical parse
bool mode_from_env() {
    const char* v = std::getenv("ORZ_MODE");
    return !(v && std::string(v) == "exact");
}

// ORZ_VERIFY=decode (bin/orz encode --verify): every finished stream goes through the library's OWN decoder
// (orz_host_decode.h, src/lz.rs:366-478) before its bytes leave the library, and must reproduce the input bit for bit;
// what the per-block validity gate (orz_verify.h) cannot see -- symbol ranking, Huffman tables, bit packing -- is covered by
// this.  The check is incremental: input and output are fed as they arrive, a chunk is decoded as soon as it is complete, so
// the streaming entry point (orz_encode) verifies a block's bytes BEFORE it hands them to the caller's sink.
bool verify_decode_on() {
    const char* v = std::getenv("ORZ_VERIFY");
    return v && std::string(v) == "decode";
}
// a whole stream in memory against its input (host or device resident)
void verify_stream_decode(const uint8_t* src, size_t n, bool src_on_device, const uint8_t* out, size_t out_len) {
    std::vector<uint8_t> host;
    if (src_on_device && n) {
        host.resize(n);
        ORZ_HIP_CHECK(hipMemcpy(host.data(), src, n, hipMemcpyDeviceToHost));
        src = host.data();
    }
    orz::host::DecodeCheck chk;
    chk.feed_input(src, n);
    chk.feed_output(out, out_len);
    chk.finish();
}

unsigned window_for(const orz::HipBackend& be, const orz_lzcfg& c, unsigned asked) {
    if (asked) return asked;
    const size_t dmax = std::max(c.match_depth, std::max(c.lazy_match_depth1, c.lazy_match_depth2));
    return be.resident_parse_waves(orz::ParseLds::make((uint32_t)dmax, false).total);
}

}  // namespace

struct orz_stream {
    std::unique_ptr<orz::HipBackend> be;
    std::unique_ptr<Enc> enc;
    orz_lzcfg cfg;
    unsigned seg, win;
    orz::ItemTrace trace;
    bool tracing = false;
    bool fast = false;
    unsigned ftile = orz::kFastTile, frounds = orz::kFastRounds;
    unsigned unit = 0;  // fast mode: bytes per unit of a block; 0 = the encoder's default (a stream that has the GPU to itself)
    double kernel_ms[4] = {0, 0, 0, 0};
    uint64_t kernel_n[4] = {0, 0, 0, 0};
    std::vector<orz::KernelRow> ktable;  // profile mode: every kernel of the last encode by name (orz_stream_get_kernel_table)
    // Build the encoder for the current settings.  The new one is constructed BEFORE the old one is dropped (a constructor
    // that throws -- bad tile size, failed allocation -- leaves the handle as it was); callers that change settings go
    // through `reconfigure`, which also restores them.
    void rebuild() {
        std::unique_ptr<Enc> fresh;
        try {
            fresh.reset(new Enc(*be, to_cfg(&cfg), seg, fast ? 64 : window_for(*be, cfg, win), fast, ftile, frounds));
        } catch (const std::exception&) {
            // a stream's state is ~5 GB: two of them may not fit a loaded device where one does.  Was the failure a bad
            // setting, the second attempt fails the same way and the handle has no encoder left (every call on it reports
            // that); was it memory, dropping the old encoder first makes room.
            if (!enc) throw;
            (void)hipGetLastError();
            enc.reset();
            be->clear_graphs();
            fresh.reset(new Enc(*be, to_cfg(&cfg), seg, fast ? 64 : window_for(*be, cfg, win), fast, ftile, frounds));
        }
        if (fast && unit && !getenv("ORZ_FAST_UNIT")) fresh->set_unit(unit);
        be->clear_graphs();  // (captured launches hold the old encoder's buffer addresses)
        enc = std::move(fresh);
        enc->trace = tracing ? &trace : nullptr;
    }
    // apply a settings change + rebuild as a transaction: on failure every setting is as before and the old encoder lives on
    template <class Change>
    void reconfigure(Change change) {
        const unsigned seg0 = seg, win0 = win, ftile0 = ftile, frounds0 = frounds;
        const bool fast0 = fast;
        change();
        try {
            rebuild();
        } catch (...) {
            seg = seg0; win = win0; ftile = ftile0; frounds = frounds0; fast = fast0;
            if (!enc) {  // (the old encoder was dropped to make room: bring it back with the old settings)
                try { rebuild(); } catch (...) {}
            }
            throw;
        }
    }
};

struct orz_lz_encoder {
    std::unique_ptr<orz::HipBackend> be;
    std::unique_ptr<Enc> enc;
    orz_lzcfg cfg{0, 0, 0};
    unsigned seg, win;
    // chunks of the block parsed last, handed out one per encode() call
    std::vector<uint8_t> framed;  // { LEB128(t) chunk[t] }*
    size_t framed_pos = 0;
    std::vector<size_t> chunk_end_spos;
    size_t next_chunk = 0;
    size_t expect_spos = 0;
    bool first_block = true;
    bool slid = false;  // forward() was called: the two bytes in front of the device window are history, not sentinel
};

extern "C" {

const char* orz_last_error(void) { return g_err.c_str(); }
const char* orz_version(void) { return "orz_hip 0.1 (bitstream: richox/orz 1.6.1)"; }

int orz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orz_lzcfg_from_level(int level, orz_lzcfg* out) {  // src/main.rs:97-102
    if (!out) return fail(ORZ_EINVAL, "null cfg");
    switch (level) {
        case 0: *out = orz_lzcfg{5, 3, 2}; return ORZ_OK;
        case 1: *out = orz_lzcfg{15, 9, 6}; return ORZ_OK;
        case 2: *out = orz_lzcfg{45, 27, 18}; return ORZ_OK;
        default: return fail(ORZ_EINVAL, "invalid level");
    }
}

void orz_free(void* p) { std::free(p); }

// ------------------------------------------------------------------------------ orz_stream
static orz_stream* stream_new(int device, const orz_lzcfg* cfg, bool lone) {
    if (!cfg_ok(cfg)) { fail(ORZ_EINVAL, "bad LZCfg"); return nullptr; }
    try {
        if (device < 0 || device >= orz_device_count()) { fail(ORZ_ENODEV, "no such HIP device"); return nullptr; }
        std::unique_ptr<orz_stream> s(new orz_stream);
        s->be.reset(new orz::HipBackend(device, lone));
        s->be->enable_arena();
        s->cfg = *cfg;
        s->seg = env_u("ORZ_SEG", kDefaultSeg);
        s->win = env_u("ORZ_WIN", 0);
        s->fast = mode_from_env();
        s->ftile = env_u("ORZ_FAST_TILE", orz::kFastTile);
        s->frounds = env_u("ORZ_FAST_ROUNDS", orz::kFastRounds);
        s->be->set_graphs(env_u("ORZ_GRAPHS", 1) != 0);
        if (!lone) s->unit = orz::kNewMax;  // one of several encoders on the device: whole blocks (orz_stream.h, unit_)
        s->rebuild();
        return s.release();
    } catch (const std::exception& e) {
        fail(ORZ_ENODEV, e.what());
        return nullptr;
    }
}
orz_stream* orz_stream_new(int device, const orz_lzcfg* cfg) { return stream_new(device, cfg, true); }
void orz_stream_free(orz_stream* s) {
    if (!s) return;
    s->enc.reset();
    s->be.reset();
    delete s;
}
int orz_stream_set_tuning(orz_stream* s, unsigned seg_bytes, unsigned window_segs) {
    if (!s) return fail(ORZ_EINVAL, "null stream");
    try {
        s->reconfigure([&] {
            if (seg_bytes) s->seg = seg_bytes;
            if (window_segs) s->win = window_segs;
        });
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}
int orz_stream_set_mode(orz_stream* s, int mode, unsigned tile_bytes, unsigned rounds) {
    if (!s || (mode != ORZ_MODE_EXACT && mode != ORZ_MODE_FAST)) return fail(ORZ_EINVAL, "bad mode");
    try {
        s->reconfigure([&] {
            s->fast = mode == ORZ_MODE_FAST;
            if (tile_bytes) s->ftile = tile_bytes;
            if (rounds) s->frounds = rounds;
        });
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}
int orz_stream_set_profile(orz_stream* s, int on) {
    if (!s) return fail(ORZ_EINVAL, "null stream");
    s->be->set_profile(on != 0);
    return ORZ_OK;
}
int orz_stream_get_kernel_times(orz_stream* s, double* ms4, uint64_t* launches4) {
    if (!s || !ms4 || !launches4) return fail(ORZ_EINVAL, "null argument");
    for (int i = 0; i < 4; i++) { ms4[i] = s->kernel_ms[i]; launches4[i] = s->kernel_n[i]; }
    return ORZ_OK;
}
long orz_stream_get_kernel_table(orz_stream* s, orz_kernel_row* rows, size_t cap) {
    if (!s) return fail(ORZ_EINVAL, "null stream");
    for (size_t i = 0; i < s->ktable.size() && i < cap && rows; i++) {
        std::memset(&rows[i], 0, sizeof rows[i]);
        std::strncpy(rows[i].name, s->ktable[i].name.c_str(), sizeof rows[i].name - 1);
        rows[i].ms = s->ktable[i].ms;
        rows[i].launches = s->ktable[i].launches;
    }
    return (long)s->ktable.size();
}
int orz_stream_get_config(orz_stream* s, orz_stream_config* out) {
    if (!s || !out) return fail(ORZ_EINVAL, "null argument");
    if (!s->enc) return fail(ORZ_ENOMEM, "the stream has no encoder (a reconfiguration ran out of device memory)");
    out->mode = s->enc->fast() ? ORZ_MODE_FAST : ORZ_MODE_EXACT;
    out->segment_bytes = s->enc->seg_size();
    out->window_segments = s->enc->window_segs();
    out->fast_tile_bytes = s->enc->fast_tile();
    out->fast_rounds = s->enc->fast_rounds();
    out->fast_row_entries = s->enc->fast_row();
    out->unit_bytes = s->enc->unit_bytes();
    return ORZ_OK;
}
// One stream through the encoder with the finished stream left in device memory (`d_dst` / `d_cap`: the caller's buffer, or nullptr
// for the encoder's own): the common body of orz_stream_encode and orz_stream_encode_to_device.  Throws.
static orz::StreamEncoder<orz::HipBackend>::DeviceResult stream_encode_device(orz_stream* s, const void* src, size_t n, int src_on_device,
                                                                              uint8_t* d_dst, size_t d_cap, orz_encode_stats* stats) {
    orz::HipBackend& be = *s->be;
    be.set_timing(stats != nullptr);
    be.begin_encode();
    struct Events {  // destroyed on every path out of this function
        orz::HipBackend& be;
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { be.set_end_event(nullptr); if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev{be};
    ORZ_HIP_CHECK(hipSetDevice(be.device()));
    ORZ_HIP_CHECK(hipEventCreate(&ev.a));
    ORZ_HIP_CHECK(hipEventCreate(&ev.b));
    ORZ_HIP_CHECK(hipEventRecord(ev.a, be.stream()));
    be.set_end_event(ev.b);  // (recorded behind the stream's last frame kernel, before the closing wait: finish_device)
    s->trace.clear();
    struct Pin {  // host input: page-lock it for the call so the block uploads are real asynchronous DMA
        void* p = nullptr;
        ~Pin() { if (p) (void)hipHostUnregister(p); }
    } pin;
    bool pinned = false;
    if (!src_on_device && n >= (1u << 20) && hipHostRegister(const_cast<void*>(src), n, hipHostRegisterDefault) == hipSuccess) {
        pin.p = const_cast<void*>(src);
        pinned = true;
    } else {
        (void)hipGetLastError();  // (registration refused: pageable copies, synchronised per block)
    }
    const auto r = orz::encode_stream_device(*s->enc, be, (const uint8_t*)src, n, src_on_device != 0, d_dst, d_cap, pinned);
    if (stats) {
        float total = 0;
        ORZ_HIP_CHECK(hipEventElapsedTime(&total, ev.a, ev.b));
        const orz::EncodeStats& st = s->enc->stats;
        stats->blocks = st.blocks; stats->sweeps = st.sweeps; stats->seg_evals = st.seg_evals;
        stats->items = st.items; stats->chunks = st.chunks; stats->in_bytes = st.in_bytes;
        stats->out_bytes = r.len;
        stats->t_prep_s = st.t_prep; stats->t_parse_s = st.t_parse; stats->t_post_s = st.t_post;
        stats->parse_kernel_ms = be.collect_timed(&stats->parse_launches, s->kernel_ms, s->kernel_n);
        stats->total_ms = total;
        stats->host_syncs = st.host_syncs + be.take_host_syncs();
        if (be.profile()) s->ktable = be.collect_named();
    }
    return r;
}
int orz_stream_encode(orz_stream* s, const void* src, size_t n, int src_on_device, uint8_t** dst, size_t* dst_len,
                      orz_encode_stats* stats) {
    if (!s || !dst || !dst_len || (!src && n)) return fail(ORZ_EINVAL, "null argument");
    if (!s->enc) return fail(ORZ_ENOMEM, "the stream has no encoder (a reconfiguration ran out of device memory)");
    try {
        const auto r = stream_encode_device(s, src, n, src_on_device, nullptr, 0, stats);
        // the finished stream comes to the host in ONE copy (round 6; before: a wait for the sizes and one for the bytes of every block)
        uint8_t* p = (uint8_t*)std::malloc(r.len ? r.len : 1);
        if (!p) return fail(ORZ_ENOMEM, "out of host memory");
        const hipError_t ce = hipMemcpy(p, r.data, r.len, hipMemcpyDeviceToHost);
        if (stats) stats->host_syncs += 1;
        if (ce != hipSuccess) { std::free(p); return fail(ORZ_ENODEV, hipGetErrorString(ce)); }
        if (verify_decode_on()) {
            try { verify_stream_decode((const uint8_t*)src, n, src_on_device != 0, p, r.len); } catch (...) { std::free(p); throw; }
        }
        *dst_len = r.len;
        *dst = p;
        return ORZ_OK;
    } catch (const std::bad_alloc&) {
        return fail(ORZ_ENOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(ORZ_ENODEV, e.what());
    }
}
size_t orz_stream_bound(size_t n) { return orz::stream_bound(n); }
int orz_stream_encode_to_device(orz_stream* s, const void* src, size_t n, int src_on_device, uint8_t* d_dst, size_t d_cap,
                                size_t* dst_len, orz_encode_stats* stats) {
    if (!s || !d_dst || !dst_len || (!src && n)) return fail(ORZ_EINVAL, "null argument");
    if (!s->enc) return fail(ORZ_ENOMEM, "the stream has no encoder (a reconfiguration ran out of device memory)");
    try {
        const auto r = stream_encode_device(s, src, n, src_on_device, d_dst, d_cap, stats);
        if (verify_decode_on()) {
            std::vector<uint8_t> host(r.len);
            ORZ_HIP_CHECK(hipMemcpy(host.data(), r.data, r.len, hipMemcpyDeviceToHost));
            verify_stream_decode((const uint8_t*)src, n, src_on_device != 0, host.data(), host.size());
        }
        *dst_len = r.len;
        return ORZ_OK;
    } catch (const std::bad_alloc&) {
        return fail(ORZ_ENOMEM, "out of host memory");
    } catch (const std::exception& e) {
        const std::string msg = e.what();
        return fail(msg.find("output buffer is too small") != std::string::npos ? ORZ_ENOMEM : ORZ_ENODEV, msg);
    }
}

int orz_stream_set_item_trace(orz_stream* s, int on) {
    if (!s || !s->enc) return fail(ORZ_EINVAL, "null stream");
    s->tracing = on != 0;
    s->enc->trace = s->tracing ? &s->trace : nullptr;
    if (!on) s->trace.clear();
    return ORZ_OK;
}
long orz_stream_get_item_trace(orz_stream* s, orz_item* out, size_t cap) {
    if (!s) return fail(ORZ_EINVAL, "null stream");
    const orz::ItemTrace& t = s->trace;
    const size_t n = t.pos.size();
    for (size_t i = 0; i < n && i < cap && out; i++) {
        orz_item it;
        it.block = t.block[i]; it.pos = t.pos[i]; it.symbol = t.sym[i]; it.rank = t.rank[i]; it.ctx = t.ctx[i];
        it.robits = t.rob[i]; it.unlikely = t.unl[i]; it.enc_len = t.enc[i]; it.after_literal = t.al[i]; it.match_len = t.mlen[i];
        it.src = t.src[i];
        out[i] = it;
    }
    return (long)n;
}

// ------------------------------------------------------------------------------ members
struct orz_members {
    std::vector<orz_stream*> workers;
    std::map<int, std::pair<uint8_t*, size_t>> arena;  // per device: where the finished members of a job are collected (orz_members_encode)
};
orz_members* orz_members_new_multi(const int* devices, int n_devices, const orz_lzcfg* cfg, int jobs_per_device) {
    if (!cfg_ok(cfg) || !devices || n_devices < 1 || n_devices > 64 || jobs_per_device < 1 || jobs_per_device > 64) {
        fail(ORZ_EINVAL, "bad argument");
        return nullptr;
    }
    std::unique_ptr<orz_members> m(new orz_members);
    auto drop = [&]() { for (orz_stream* w : m->workers) orz_stream_free(w); m->workers.clear(); };
    for (int d = 0; d < n_devices; d++)
        for (int i = 0; i < jobs_per_device; i++) {
            orz_stream* s = stream_new(devices[d], cfg, n_devices * jobs_per_device == 1);
            if (!s) { drop(); return nullptr; }
            if (jobs_per_device > 1 && !s->fast) {  // exact mode: several streams share the GPU: a smaller speculative window each
                s->win = env_u("ORZ_MEMBER_WIN", std::max(256u, window_for(*s->be, *cfg, 0) / (unsigned)jobs_per_device));
                try { s->reconfigure([] {}); } catch (const std::exception& e) { fail(ORZ_ENODEV, e.what()); orz_stream_free(s); drop(); return nullptr; }
            }
            m->workers.push_back(s);
        }
    return m.release();
}
orz_members* orz_members_new(int device, const orz_lzcfg* cfg, int jobs) { return orz_members_new_multi(&device, 1, cfg, jobs); }
void orz_members_free(orz_members* m) {
    if (!m) return;
    for (orz_stream* w : m->workers) orz_stream_free(w);
    for (auto& kv : m->arena)
        if (kv.second.first) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.first); }
    delete m;
}
// The members of a job on the workers' devices.  A worker leaves a finished member in its encoder's own device buffer
// (encode_stream_device: one host wait per block, one per member) and moves it -- device to device, on its copy stream, no host
// wait: the next member's frame kernels queue behind the copy -- into the job's ARENA on that device at an offset drawn from an
// atomic counter; `place[k]` says where member k lies.  Arena = the caller's buffer (orz_members_encode_to_device) or one the
// members object owns; a member that does not fit an OWNED arena goes to host memory instead (incompressible input: the arena
// is sized for ratio 0.5), into the caller's it fails the job.
namespace {
struct MemberPlace { int device = -1; size_t off = 0, len = 0; std::vector<uint8_t> host; };
struct MemberArena {
    int device; uint8_t* p; size_t cap; bool caller;
    std::atomic<size_t> used{0};
    MemberArena(int d, uint8_t* q, size_t c, bool cl) : device(d), p(q), cap(c), caller(cl) {}
};
int members_run(orz_members* m, const void* src, size_t n, int src_on_device, size_t member_bytes, std::vector<std::unique_ptr<MemberArena>>& arenas,
                std::vector<MemberPlace>& place) {
    const size_t nm = place.size();
    std::atomic<size_t> next{0};
    std::atomic<int> rc{ORZ_OK};
    std::string err;
    const bool verify = verify_decode_on();
    auto arena_of = [&](int dev) -> MemberArena* {
        for (auto& a : arenas) if (a->device == dev) return a.get();
        return nullptr;
    };
    auto work = [&](orz_stream* s) {
        try {
            orz::HipBackend& be = *s->be;
            ORZ_HIP_CHECK(hipSetDevice(be.device()));  // every host thread talks to its worker's device
            MemberArena* ar = arena_of(be.device());
            bool copied = false;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nm || rc.load() != ORZ_OK) break;
                if (!s->enc) throw std::runtime_error("a worker has no encoder (a reconfiguration ran out of device memory)");
                const size_t off = i * member_bytes, len = n == 0 ? 0 : std::min(member_bytes, n - off);
                be.begin_encode();
                const auto r = orz::encode_stream_device(*s->enc, be, (const uint8_t*)src + off, len, src_on_device != 0, nullptr, 0);
                MemberPlace& pl = place[i];
                pl.len = r.len;
                const size_t at = ar ? ar->used.fetch_add(r.len) : 0;
                if (ar && at + r.len <= ar->cap) {
                    pl.device = be.device(); pl.off = at;
                    be.select(3);
                    be.d2d(ar->p + at, r.data, r.len);
                    be.select(0);
                    copied = true;
                } else {
                    if (ar && ar->caller) throw std::runtime_error("the output buffer is too small for the members' streams");
                    pl.host.resize(r.len);
                    ORZ_HIP_CHECK(hipMemcpy(pl.host.data(), r.data, r.len, hipMemcpyDeviceToHost));
                }
                if (verify) {
                    std::vector<uint8_t> h(r.len);
                    ORZ_HIP_CHECK(hipMemcpy(h.data(), r.data, r.len, hipMemcpyDeviceToHost));
                    verify_stream_decode((const uint8_t*)src + off, len, src_on_device != 0, h.data(), h.size());
                }
            }
            if (copied) { be.select(3); be.sync(); be.select(0); }  // (the worker's moves into the arena: one wait per job)
        } catch (const std::exception& e) {
            int expect = ORZ_OK;
            const std::string msg = e.what();
            if (rc.compare_exchange_strong(expect, msg.find("output buffer is too small") != std::string::npos ? ORZ_ENOMEM : ORZ_ENODEV)) err = msg;
        }
    };
    struct Joiner {  // joins whatever was started, also when starting a later thread throws
        std::vector<std::thread> th;
        ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } joiner;
    for (size_t i = 1; i < m->workers.size(); i++) joiner.th.emplace_back(work, m->workers[i]);
    work(m->workers[0]);
    for (auto& t : joiner.th) t.join();
    if (rc.load() != ORZ_OK) return fail(rc.load(), err);
    return ORZ_OK;
}
}  // namespace
int orz_members_encode(orz_members* m, const void* src, size_t n, int src_on_device, size_t member_bytes, uint8_t** dst,
                       size_t* dst_len, size_t* n_members_out) {
    if (!m || !dst || !dst_len || (!src && n) || member_bytes == 0) return fail(ORZ_EINVAL, "bad argument");
    if (src_on_device && m->workers.size() > 1) {
        const int d0 = m->workers[0]->be->device();
        for (orz_stream* w : m->workers)
            if (w->be->device() != d0) return fail(ORZ_EINVAL, "device-resident input needs all workers on that device");
    }
    try {
        const size_t nm = n == 0 ? 1 : (n + member_bytes - 1) / member_bytes;
        // an arena per device the workers sit on, owned by the members object and kept between calls: sized for ratio 0.5 (text:
        // 0.28) or the streams' bound, whichever is smaller
        std::vector<std::unique_ptr<MemberArena>> arenas;
        const size_t want = std::min(nm * orz::stream_bound(std::min(member_bytes, n ? n : 1)), n / 2 + nm * 65536 + (32u << 20));
        for (orz_stream* w : m->workers) {
            const int dev = w->be->device();
            bool have = false;
            for (auto& a : arenas) have = have || a->device == dev;
            if (have) continue;
            auto& slot = m->arena[dev];
            if (slot.second < want) {
                ORZ_HIP_CHECK(hipSetDevice(dev));
                if (slot.first) (void)hipFree(slot.first);
                slot = {nullptr, 0};
                void* p = nullptr;
                if (hipMalloc(&p, want) == hipSuccess) slot = {(uint8_t*)p, want};
                else (void)hipGetLastError();  // (no arena on this device: its members go through host memory)
            }
            if (slot.first) {
                std::unique_ptr<MemberArena> a(new MemberArena(dev, slot.first, slot.second, false));
                arenas.push_back(std::move(a));
            }
        }
        std::vector<MemberPlace> place(nm);
        const int rc = members_run(m, src, n, src_on_device, member_bytes, arenas, place);
        if (rc != ORZ_OK) return rc;
        // the members in input order: ONE copy each, from where it lies to its place in the result (round 6; before: device ->
        // a vector per member -> the result)
        size_t total = 0;
        for (auto& pl : place) total += pl.len;
        uint8_t* p = (uint8_t*)std::malloc(total ? total : 1);
        if (!p) return fail(ORZ_ENOMEM, "malloc failed");
        size_t at = 0;
        int cur = -1;
        for (auto& pl : place) {
            if (pl.device >= 0) {
                if (cur != pl.device) { (void)hipSetDevice(pl.device); cur = pl.device; }
                uint8_t* base = nullptr;
                for (auto& a : arenas) if (a->device == pl.device) base = a->p;
                const hipError_t ce = hipMemcpy(p + at, base + pl.off, pl.len, hipMemcpyDeviceToHost);
                if (ce != hipSuccess) { std::free(p); return fail(ORZ_ENODEV, hipGetErrorString(ce)); }
            } else {
                std::memcpy(p + at, pl.host.data(), pl.len);
            }
            at += pl.len;
        }
        *dst = p;
        *dst_len = total;
        if (n_members_out) *n_members_out = nm;
        return ORZ_OK;
    } catch (const std::exception& e) {  // (allocation / thread start failures never cross the C boundary)
        return fail(ORZ_ENOMEM, e.what());
    }
}
int orz_members_encode_to_device(orz_members* m, const void* src, size_t n, int src_on_device, size_t member_bytes, uint8_t* d_dst,
                                 size_t d_cap, size_t* offs, size_t* lens, size_t* n_members_out) {
    if (!m || !d_dst || !offs || !lens || (!src && n) || member_bytes == 0) return fail(ORZ_EINVAL, "bad argument");
    const int d0 = m->workers[0]->be->device();
    for (orz_stream* w : m->workers)
        if (w->be->device() != d0) return fail(ORZ_EINVAL, "device-resident output needs all workers on one device");
    try {
        const size_t nm = n == 0 ? 1 : (n + member_bytes - 1) / member_bytes;
        std::vector<std::unique_ptr<MemberArena>> arenas;
        arenas.emplace_back(new MemberArena(d0, d_dst, d_cap, true));
        std::vector<MemberPlace> place(nm);
        const int rc = members_run(m, src, n, src_on_device, member_bytes, arenas, place);
        if (rc != ORZ_OK) return rc;
        for (size_t k = 0; k < nm; k++) { offs[k] = place[k].off; lens[k] = place[k].len; }
        if (n_members_out) *n_members_out = nm;
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_ENOMEM, e.what());
    }
}
int orz_decode_members_mem(const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len, size_t* n_members_out) {
    if ((!src && n) || !dst || !dst_len) return fail(ORZ_EINVAL, "bad argument");
    try {
        std::vector<uint8_t> out;
        size_t at = 0, members = 0;
        orz::host::DecodeWorkspace ws;  // one window / model allocation for all members
        while (at < n) {
            orz::host::decode_stream(
                ws,
                [&](uint8_t* buf, size_t k) {
                    if (at + k > n) return false;
                    std::memcpy(buf, src + at, k);
                    at += k;
                    return true;
                },
                [&](const uint8_t* buf, size_t k) { out.insert(out.end(), buf, buf + k); }, [](bool, size_t, size_t) {});
            members++;
        }
        uint8_t* p = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        if (!p) return fail(ORZ_ENOMEM, "malloc failed");
        std::memcpy(p, out.data(), out.size());
        *dst = p;
        *dst_len = out.size();
        if (n_members_out) *n_members_out = members;
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}

int orz_decode_members_device(int device, const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len,
                              size_t* n_members_out, orz_decode_stats* stats) {
    if ((!src && n) || !dst || !dst_len) return fail(ORZ_EINVAL, "bad argument");
    if (device < 0 || device >= orz_device_count()) return fail(ORZ_ENODEV, "no such HIP device");
    try {
        orz::HipBackend be(device);
        std::vector<uint8_t> out;
        orz::DecodeStats st;
        orz::decode_members_device(be, src, n, out, st, env_u("ORZ_DECODE_SLOTS", 2048));
        uint8_t* p = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        if (!p) return fail(ORZ_ENOMEM, "malloc failed");
        std::memcpy(p, out.data(), out.size());
        *dst = p;
        *dst_len = out.size();
        if (n_members_out) *n_members_out = (size_t)st.members;
        if (stats) {
            stats->members = st.members; stats->in_bytes = st.in_bytes; stats->out_bytes = st.out_bytes;
            stats->launches = st.launches; stats->kernel_ms = st.kernel_ms; stats->total_s = st.total_s;
        }
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}

// ------------------------------------------------------------------------------ Huffman tables alone
size_t orz_huffman_stride(void) { return orz::kHwStride; }
int orz_huffman_tables(int device, const uint32_t* weights, size_t nchunks, uint8_t* lens, uint16_t* codes, double* elapsed_us) {
    if (!weights || !lens || !codes || nchunks > (1u << 20)) return fail(ORZ_EINVAL, "bad argument");
    if (device < 0 || device >= orz_device_count()) return fail(ORZ_ENODEV, "no such HIP device");
    const size_t n = nchunks * orz::kHwStride;
    for (size_t i = 0; i < n; i++)
        if (weights[i] > orz::HuffBuild::kMaxWeight) return fail(ORZ_EINVAL, "symbol weight of 2^23 or more");
    if (!n) return ORZ_OK;
    try {
        orz::HipBackend be(device);
        struct Bufs {
            orz::HipBackend& be;
            uint32_t* w = nullptr; uint8_t* l = nullptr; uint16_t* c = nullptr;
            ~Bufs() { be.free(w); be.free(l); be.free(c); }
        } b{be};
        b.w = be.alloc<uint32_t>(n, false);
        b.l = be.alloc<uint8_t>(n);
        b.c = be.alloc<uint16_t>(n);
        be.h2d(b.w, weights, n * 4);
        const orz::HuffBuild f{b.w, (uint32_t)nchunks, b.l, b.c};
        be.huffbuild(f);  // (warm: code object load)
        struct Events {  // destroyed on every path out
            hipEvent_t a = nullptr, b = nullptr;
            ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
        } ev;
        ORZ_HIP_CHECK(hipEventCreate(&ev.a));
        ORZ_HIP_CHECK(hipEventCreate(&ev.b));
        ORZ_HIP_CHECK(hipEventRecord(ev.a, be.stream()));
        be.huffbuild(f);
        ORZ_HIP_CHECK(hipEventRecord(ev.b, be.stream()));
        be.d2h(lens, b.l, n);
        be.d2h(codes, b.c, n * 2);
        float ms = 0;
        ORZ_HIP_CHECK(hipEventElapsedTime(&ms, ev.a, ev.b));
        if (elapsed_us) *elapsed_us = (double)ms * 1000.0;
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}

// ------------------------------------------------------------------------------ orz_lz_encoder
orz_lz_encoder* orz_lz_encoder_new(int device) {
    try {
        if (device < 0 || device >= orz_device_count()) { fail(ORZ_ENODEV, "no such HIP device"); return nullptr; }
        std::unique_ptr<orz_lz_encoder> e(new orz_lz_encoder);
        e->be.reset(new orz::HipBackend(device));
        e->be->enable_arena();
        e->seg = env_u("ORZ_SEG", kDefaultSeg);
        e->win = env_u("ORZ_WIN", 0);
        return e.release();
    } catch (const std::exception& ex) {
        fail(ORZ_ENODEV, ex.what());
        return nullptr;
    }
}
void orz_lz_encoder_free(orz_lz_encoder* e) {
    if (!e) return;
    e->enc.reset();
    e->be.reset();
    delete e;
}

int orz_lz_encoder_encode(orz_lz_encoder* e, const orz_lzcfg* cfg, const uint8_t* sbuf, size_t sbuf_len, uint8_t* tbuf,
                          size_t tbuf_cap, size_t spos, size_t* spos_out, size_t* tlen_out) {
    if (!e || !sbuf || !tbuf || !spos_out || !tlen_out || !cfg_ok(cfg)) return fail(ORZ_EINVAL, "bad argument");
    if (sbuf_len > orz::kBlock || spos > sbuf_len || spos < orz::kPre) return fail(ORZ_EINVAL, "bad window geometry");
    try {
        const bool same_cfg = e->enc && std::memcmp(&e->cfg, cfg, sizeof *cfg) == 0;
        if (!e->enc) {
            e->cfg = *cfg;
            const bool fast = mode_from_env();
            e->enc.reset(new Enc(*e->be, to_cfg(cfg), e->seg, fast ? 64 : window_for(*e->be, *cfg, e->win), fast,
                                 env_u("ORZ_FAST_TILE", orz::kFastTile), env_u("ORZ_FAST_ROUNDS", orz::kFastRounds)));
        } else if (!same_cfg) {
            return fail(ORZ_EINVAL, "LZCfg changed inside a stream");
        }
        const bool continuing = e->next_chunk < e->chunk_end_spos.size() && spos == e->expect_spos;
        if (!continuing) {
            if (spos != orz::kPre) return fail(ORZ_EINVAL, "a block must start at SBVEC_PREMATCH_LEN");
            // upload the caller's window exactly as it is, sentinel pads included (src/lib.rs:67-69) -- except, after a
            // forward(), the pad in FRONT of the window: the caller's copy_within (src/lib.rs:83) leaves the sentinel's zeros
            // there, but the context of the item start at window offset 1 (hash1 of offset 0, src/lz.rs:482-486) looks at the
            // byte before the window, and this encoder rebuilds its tables from the window's bytes -- slide(false) kept the
            // two real bytes on the device, and the upload must not wipe them (round 3's slide defect, at this seam)
            if (e->slid) e->be->h2d(e->enc->dwin(), sbuf, (size_t)sbuf_len + orz::kSent);
            else e->be->h2d(e->enc->dwinbuf(), sbuf - orz::kSent, (size_t)sbuf_len + 2 * orz::kSent);
            e->framed.clear();
            e->chunk_end_spos.clear();
            e->enc->encode_block((uint32_t)(sbuf_len - orz::kPre), e->framed, &e->chunk_end_spos);
            e->framed_pos = 0;
            e->next_chunk = 0;
        }
        // un-frame the next chunk: LEB128(t) chunk[t]
        size_t t = 0, shift = 0, at = e->framed_pos;
        for (;;) {
            uint8_t b = e->framed[at++];
            t |= (size_t)(b & 0x7f) << shift;
            shift += 7;
            if (!(b & 0x80)) break;
        }
        if (t > tbuf_cap) return fail(ORZ_ENOMEM, "tbuf too small");
        std::memcpy(tbuf, e->framed.data() + at, t);
        e->framed_pos = at + t;
        *tlen_out = t;
        *spos_out = e->chunk_end_spos[e->next_chunk++];
        e->expect_spos = *spos_out;
        return ORZ_OK;
    } catch (const std::exception& ex) {
        return fail(ORZ_ENODEV, ex.what());
    }
}

int orz_lz_encoder_forward(orz_lz_encoder* e, size_t forward_len) {
    if (!e || !e->enc) return fail(ORZ_EINVAL, "forward before encode");
    if (forward_len != orz::kNewMax) return fail(ORZ_EINVAL, "forward_len must be 2^24");
    try {
        e->enc->slide(false);  // the caller re-supplies the slid window on the next encode()
        e->slid = true;
        e->chunk_end_spos.clear();
        e->next_chunk = 0;
        return ORZ_OK;
    } catch (const std::exception& ex) {
        return fail(ORZ_ENODEV, ex.what());
    }
}

// ------------------------------------------------------------------------------ decode (host)
struct orz_lz_decoder {
    orz::host::Decoder dec;
};
orz_lz_decoder* orz_lz_decoder_new(void) {
    try {
        return new orz_lz_decoder();
    } catch (const std::exception& e) {
        fail(ORZ_ENOMEM, e.what());
        return nullptr;
    }
}
void orz_lz_decoder_free(orz_lz_decoder* d) { delete d; }
int orz_lz_decoder_decode(orz_lz_decoder* d, const uint8_t* tbuf, size_t tlen, uint8_t* sbuf, size_t spos,
                          size_t* spos_end_out) {
    if (!d || !tbuf || !sbuf || !spos_end_out || spos < 3) return fail(ORZ_EINVAL, "bad argument");
    try {
        *spos_end_out = d->dec.decode(tbuf, tlen, sbuf, spos);
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}
int orz_lz_decoder_forward(orz_lz_decoder* d, size_t forward_len) {
    if (!d) return fail(ORZ_EINVAL, "null decoder");
    d->dec.forward(forward_len);
    return ORZ_OK;
}
int orz_decode(orz_read_fn rd, void* rctx, orz_write_fn wr, void* wctx, orz_progress_fn prog, void* pctx) {
    if (!rd || !wr) return fail(ORZ_EINVAL, "bad argument");
    int io = ORZ_OK;
    try {
        orz::host::decode_stream(
            [&](uint8_t* buf, size_t n) {
                size_t got = 0;
                while (got < n) {
                    ssize_t r = rd(rctx, buf + got, n - got);
                    if (r < 0) { io = ORZ_EIO; return false; }
                    if (r == 0) return false;
                    got += (size_t)r;
                }
                return true;
            },
            [&](const uint8_t* buf, size_t n) {
                if (n && wr(wctx, buf, n) != 0) { io = ORZ_EIO; throw std::runtime_error("write failed"); }
            },
            [&](bool fin, size_t a, size_t b) {
                if (prog) prog(pctx, fin ? 1 : 0, a, b);
            });
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(io != ORZ_OK ? io : ORZ_EINVAL, e.what());
    }
}
int orz_decode_mem(const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len, size_t* consumed) {
    if ((!src && n) || !dst || !dst_len) return fail(ORZ_EINVAL, "bad argument");
    try {
        std::vector<uint8_t> out;
        size_t at = 0;
        orz::host::decode_stream(
            [&](uint8_t* buf, size_t k) {
                if (at + k > n) return false;
                std::memcpy(buf, src + at, k);
                at += k;
                return true;
            },
            [&](const uint8_t* buf, size_t k) { out.insert(out.end(), buf, buf + k); }, [](bool, size_t, size_t) {});
        uint8_t* p = (uint8_t*)std::malloc(out.size() ? out.size() : 1);
        if (!p) return fail(ORZ_ENOMEM, "malloc failed");
        std::memcpy(p, out.data(), out.size());
        *dst = p;
        *dst_len = out.size();
        if (consumed) *consumed = at;
        return ORZ_OK;
    } catch (const std::exception& e) {
        return fail(ORZ_EINVAL, e.what());
    }
}

// ------------------------------------------------------------------------------ orz::encode
int orz_encode(orz_read_fn rd, void* rctx, orz_write_fn wr, void* wctx, const orz_lzcfg* cfg, orz_progress_fn prog,
               void* pctx, int device) {
    if (!rd || !wr || !cfg_ok(cfg)) return fail(ORZ_EINVAL, "bad argument");
    orz_stream* s = orz_stream_new(device, cfg);
    if (!s) return ORZ_ENODEV;
    int rc = ORZ_OK;
    try {
        Enc& enc = *s->enc;
        orz::HipBackend& be = *s->be;
        enc.reset();
        struct PinnedBuf {  // the block read buffer, page-locked (pinned hipMemcpyAsync feed)
            uint8_t* p = nullptr;
            PinnedBuf() { if (hipHostMalloc((void**)&p, orz::kNewMax, hipHostMallocDefault) != hipSuccess) { p = nullptr; throw std::runtime_error("hipHostMalloc failed"); } }
            ~PinnedBuf() { if (p) (void)hipHostFree(p); }
            uint8_t* data() { return p; }
            size_t size() const { return orz::kNewMax; }
        } in;
        std::vector<uint8_t> out;
        size_t in_total = 0, out_total = 0;
        bool first = true;
        std::unique_ptr<orz::host::DecodeCheck> chk(verify_decode_on() ? new orz::host::DecodeCheck : nullptr);  // (verifies a block's bytes before the sink sees them)
        for (;;) {
            // read_repeatedly, src/lib.rs:42-52: fill the block or hit EOF
            size_t got = 0;
            while (got < in.size()) {
                ssize_t r = rd(rctx, in.data() + got, in.size() - got);
                if (r < 0) { rc = fail(ORZ_EIO, "read failed"); break; }
                if (r == 0) break;
                got += (size_t)r;
            }
            if (rc != ORZ_OK || got == 0) break;
            if (!first) enc.slide();
            first = false;
            be.h2d_pinned(enc.dwin() + orz::kPre, in.data(), got);  // encode_block syncs before `in` is refilled
            if (chk) chk->feed_input(in.data(), got);
            out.clear();
            enc.encode_block_units((uint32_t)got, in_total == 0 && got == in.size(), out);
            if (chk && !out.empty()) chk->feed_output(out.data(), out.size());
            if (!out.empty() && wr(wctx, out.data(), out.size()) != 0) { rc = fail(ORZ_EIO, "write failed"); break; }
            in_total += got;
            out_total += out.size();
            if (prog) prog(pctx, 0, in_total, out_total);
            if (got < in.size()) break;
        }
        if (rc == ORZ_OK) {  // the last block's tail stage is still in flight: take its bytes
            out.clear();
            enc.finish(out);
            if (chk) {
                const uint8_t z = 0;
                if (!out.empty()) chk->feed_output(out.data(), out.size());
                chk->feed_output(&z, 1);
                chk->finish();
            }
            if (!out.empty() && wr(wctx, out.data(), out.size()) != 0) rc = fail(ORZ_EIO, "write failed");
            out_total += out.size();
        }
        if (rc == ORZ_OK) {
            const uint8_t eof = 0;  // write_len(0), src/lib.rs:89
            if (wr(wctx, &eof, 1) != 0) rc = fail(ORZ_EIO, "write failed");
            out_total += 1;
            if (prog) prog(pctx, 1, in_total, out_total);
        }
    } catch (const std::exception& ex) {
        rc = fail(ORZ_ENODEV, ex.what());
    }
    orz_stream_free(s);
    return rc;
}

}  // extern "C"
