// backend_hip.h -- the product backend: runs the kernel bodies of orz_kernels.h as HIP kernels on
// gfx950 (MI355X).  One HipBackend = one device + one HIP stream; every launch, copy and memset of
// a stream encoder is ordered on that stream (so HIP events recorded on it bracket exactly the
// encoder's device work).
//
// Library primitives used as plumbing: rocPRIM device radix sort / exclusive scan (the candidate
// list build).  Everything on the orz path itself is a hand-written kernel body.
#pragma once
#include <hip/hip_runtime.h>

#include <cxxabi.h>

#include <chrono>
#include <condition_variable>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <rocprim/rocprim.hpp>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "orz_kernels.h"
#include "orz_parse.h"
#include "orz_symrank.h"

namespace orz {

#define ORZ_HIP_CHECK(expr)                                                                              \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
    } while (0)

template <class F>
__global__ __launch_bounds__(256) void orz_thread_kernel(F f, size_t n) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < n) f(tid);
}

// The same with an occupancy target (waves per SIMD) for the register allocator: FastEval runs at 82 registers = 5 waves
// per SIMD when left alone, and one launch of it oversubscribes every wave slot of the GPU -- more slots are
// more loads in flight.  ORZ_EVAL_WAVES at build time (tools/dev: 6 / 7 / 8 tried in round 4).
#if defined(ORZ_EVAL_WAVES)
template <class F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ORZ_EVAL_WAVES, ORZ_EVAL_WAVES))) void orz_thread_kernel_occ(F f, size_t n) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < n) f(tid);
}
#endif

// wave-cooperative kernels: one 64-lane wavefront per block, dynamic LDS
struct DevWave {
    uint8_t* lds_;
    __device__ __forceinline__ uint32_t lane() const { return threadIdx.x; }
    __device__ __forceinline__ uint32_t block() const { return blockIdx.x; }
    __device__ __forceinline__ uint8_t* lds() const { return lds_; }
    __device__ __forceinline__ uint64_t ballot(bool p) const { return __ballot(p); }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // value of lane `src` (wave-uniform index) in every lane: v_readlane_b32
    __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src) const {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
    }
    // value of lane `src` (any per-lane index): ds_bpermute
    __device__ __forceinline__ uint32_t shfl(uint32_t v, uint32_t src) const { return (uint32_t)__shfl((int)v, (int)src, 64); }
    __device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t src) const {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
    __device__ __forceinline__ unsigned long long clock() const { return __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ unsigned long long wallclock() const { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz, chip-wide
};
template <class K>
__global__ __launch_bounds__(64) void orz_wave_kernel(K k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t orz_dyn_lds[];
    DevWave w{orz_dyn_lds};
    k(w);
}

// kernels of ONE workgroup that run K::kPhases phases with a barrier between them, dynamic LDS
template <class K>
__global__ __launch_bounds__(1024) void orz_group_kernel(K k, bool use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint8_t orz_dyn_lds[];
    for (uint32_t ph = 0; ph < K::kPhases; ph++) {
        if (ph) {
            __threadfence_block();
            __syncthreads();
        }
        k.phase(ph, threadIdx.x, blockDim.x, orz_dyn_lds, use_lds);
    }
}

// ring ordinals after a sweep (orz_parse.h): block = chunk of kRankChunk segments, thread = ctx
__global__ __launch_bounds__(256) void orz_rank_kernel(RankArgs a, uint32_t nchunks, uint32_t nvblk) {
    __shared__ uint32_t rows[(kRankChunk + 1) * 256];
    auto sync = [] __device__() { __syncthreads(); };
    if (blockIdx.x < nchunks) rank_chunk(a, blockIdx.x, threadIdx.x, rows, sync);
    else if (blockIdx.x < nchunks + nvblk)  // the other blocks refresh the bitmap summaries for the next sweep
        rebuild_summaries(a.vbits, a.v1, a.v2, a.nvwords, blockIdx.x - nchunks, threadIdx.x, (uint64_t*)rows, sync);
    else
        rebuild_summaries(a.kbits, a.k1, a.k2, a.nkwords, blockIdx.x - nchunks - nvblk, threadIdx.x, (uint64_t*)rows, sync);
}

// (the symbol-ranking chains: orz_symrank.h)

struct FastEval;  // (orz_fast.h: the one kernel with an occupancy target, see orz_thread_kernel_occ)

// How many encoders of a process may have a block's parse in flight on a device at a time (ORZ_PARSE_TOKENS; 0 = no limit).
// The parse kernels of different encoders do not fill each other's gaps, they stretch each other: eight
// encoders whose main streams happened to share hardware queues in pairs ran at 530 MB/s, the same eight on queues of their
// own at 410.  The token makes that an explicit, deterministic policy instead of an accident of the runtime's queue
// assignment: an encoder takes one before it queues a block's prep and gives it back when the block's items are handed to
// the ranking chain; symbol ranking, Huffman, packing and copies of the other encoders run beside it as before.
class ParseTokens {
   public:
    static ParseTokens& of(int device) {
        static ParseTokens t[64];
        return t[device & 63];
    }
    void acquire() {
        if (!limit()) return;
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return busy_ < limit(); });
        busy_++;
    }
    void release() {
        if (!limit()) return;
        { std::lock_guard<std::mutex> lk(m_); busy_--; }
        cv_.notify_one();
    }
    static int limit() {
        static const int n = getenv("ORZ_PARSE_TOKENS") ? atoi(getenv("ORZ_PARSE_TOKENS")) : 0;
        return n;
    }

   private:
    std::mutex m_;
    std::condition_variable cv_;
    int busy_ = 0;
};

// The HIP streams of the encoders of a process come from a pool and go back to it.  The runtime maps streams onto its
// hardware queues (GPU_MAX_HW_QUEUES, 16 here) when they are CREATED, and how the encoders' streams share queues decides
// how their kernels overlap: a process's first eight encoders (32 streams created in a row: main streams share queues in
// pairs, ranking streams in pairs, ...) encode at 530 MB/s, every later set of eight -- created after others were destroyed, or
// simply later -- at 410, with the same clocks, memory and code (round 4: tools/dev/members_sets.py; 24 queues: 470 / 455 /
// 460, 32 queues: 413 for every set).  Reusing the streams keeps every set on the first set's mapping.
class StreamPool {
   public:
    static constexpr int kStreams = 4;
    struct Set { hipStream_t s[kStreams] = {nullptr, nullptr, nullptr, nullptr}; };
    static StreamPool& get() {
        static StreamPool* p = new StreamPool;  // (never destroyed: streams outlive every static of the runtime's clients)
        return *p;
    }
    // The first request for a (device, priority) kind creates kBlockSets sets in one go -- 32 streams in a row, however the
    // process goes on to create its encoders one, two, four or eight at a time: every encoder of the first eight runs on
    // the mapping a process's first eight encoders get (the good one, see above).  Later requests take what is free.
    static constexpr int kBlockSets = 8;
    // (`block` false: a lone encoder -- bin/orz, bench.py -- takes one set and does not pay for 32 stream creations, ~300 ms
    // of a process's start)
    Set take(int device, bool rank_prio, bool block) {
        std::lock_guard<std::mutex> lk(m_);
        auto& v = free_[{device, rank_prio}];
        if (v.empty()) {
            const int n = (!block || blocks_[{device, rank_prio}]) ? 1 : kBlockSets;
            if (n > 1) blocks_[{device, rank_prio}] = 1;
            std::vector<Set> fresh;
            for (int k = 0; k < n; k++) fresh.push_back(create(rank_prio));
            made_[{device, rank_prio}] += n;
            for (int k = n; k-- > 0;) v.push_back(fresh[k]);  // (handed out in the order they were created)
        }
        Set t = v.back();
        v.pop_back();
        return t;
    }
    void give(int device, bool rank_prio, const Set& t) {
        std::lock_guard<std::mutex> lk(m_);
        free_[{device, rank_prio}].push_back(t);
    }

   private:
    // (Round 5 measured two other ways to share the GPU here -- a share of the compute units per encoder through
    // hipExtStreamCreateWithCUMask, and several encoders on one main stream -- both worse; the knobs are gone.)
    static Set create(bool rank_prio) {
        Set t;
        int prio_low = 0, prio_high = 0;
        ORZ_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        for (int i = 0; i < kStreams; i++) {
            if (i == 1 && rank_prio) ORZ_HIP_CHECK(hipStreamCreateWithPriority(&t.s[i], hipStreamNonBlocking, prio_high));
            else ORZ_HIP_CHECK(hipStreamCreateWithFlags(&t.s[i], hipStreamNonBlocking));
        }
        return t;
    }
    std::mutex m_;
    std::map<std::pair<int, bool>, std::vector<Set>> free_;
    std::map<std::pair<int, bool>, int> made_, blocks_;
};

// Names of the kernels a profiled encode brackets (profile mode, bench.py's roofline leg): one id per kernel type of the
// process, by the functor's type name; the library calls (sorts, scans, fills, copies) get ids of their own.
class KernelNames {
   public:
    static KernelNames& get() {
        static KernelNames k;
        return k;
    }
    int id_of(const std::string& name) {
        std::lock_guard<std::mutex> lk(m_);
        for (size_t i = 0; i < names_.size(); i++)
            if (names_[i] == name) return (int)i;
        names_.push_back(name);
        return (int)names_.size() - 1;
    }
    std::string name(int id) {
        std::lock_guard<std::mutex> lk(m_);
        return id >= 0 && (size_t)id < names_.size() ? names_[id] : std::string("?");
    }
    template <class F>
    static int of() {
        static const int id = get().id_of(pretty(typeid(F).name()));
        return id;
    }
    static int lib(const char* what) { return get().id_of(what); }

   private:
    static std::string pretty(const char* mangled) {
        int st = 0;
        char* d = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
        std::string r = st == 0 && d ? d : mangled;
        std::free(d);
        if (r.rfind("orz::", 0) == 0) r = r.substr(5);
        return r;
    }
    std::mutex m_;
    std::vector<std::string> names_;
};
struct KernelRow {
    std::string name;
    double ms = 0;
    uint64_t launches = 0;
};

class HipBackend {
   public:
    // `lone`: the encoder has the GPU to itself (a single stream: bench.py, orz_stream_new, bin/orz without --jobs); the
    // workers of a members job are not lone
    explicit HipBackend(int device, bool lone = true) : device_(device), lone_(lone) {
        ORZ_HIP_CHECK(hipSetDevice(device_));
        // Stream 1 carries nothing but the symbol-ranking launches -- the serial chain of an orz stream.  A lone encoder
        // gives it the device's highest priority so that a ranking launch does not queue behind the dispatch of the next
        // block's parse grids; the workers of a members job leave it at the default.  Neither choice moved a measurement
        // beyond the run-to-run spread (bench 298...303 MB/s, eight encoders 454...557).  ORZ_RANK_PRIO=0/1 overrides.
        const char* rp = getenv("ORZ_RANK_PRIO");
        rank_prio_ = rp ? atoi(rp) != 0 : false;  // (round 4: no priority stream unless asked for -- it never moved a measurement, and
                                                  // one kind of stream set keeps every encoder of a process on the pool's first block)
        (void)lone;
        {
            const StreamPool::Set t = StreamPool::get().take(device_, rank_prio_, !lone);
            for (int i = 0; i < kStreams; i++) streams_[i] = t.s[i];
        }
        stream_ = streams_[0];
        for (int i = 0; i < kEvents; i++) ORZ_HIP_CHECK(hipEventCreateWithFlags(&sev_[i], hipEventDisableTiming));
        // temp storage big enough for the largest sort / scan of a block
        size_t s1 = 0, s2 = 0;
        uint64_t* k = nullptr;
        uint32_t* u = nullptr;
        ORZ_HIP_CHECK(rocprim::radix_sort_keys(nullptr, s1, k, k, (size_t)kWLen, 0, 64, stream_));
        ORZ_HIP_CHECK(rocprim::exclusive_scan(nullptr, s2, u, u, 0u, (size_t)kWLen, rocprim::plus<uint32_t>(), stream_));
        size_t s3 = 0;
        ORZ_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, s3, u, u, u, u, (size_t)kWLen, 0, 32, stream_));
        if (s3 > s1) s1 = s3;
        size_t s4 = 0;
        ORZ_HIP_CHECK(rocprim::inclusive_scan(nullptr, s4, u, u, (size_t)kWLen, rocprim::maximum<uint32_t>(), stream_));
        if (s4 > s2) s2 = s4;
        // The sorts run on the main stream only: its work area is the large one (a second copy of the keys and values of the
        // biggest sort).  Streams 1 and 2 carry the ranking launches and the tail stage, whose only library call is a scan over
        // the items: their work areas hold a scan's few hundred kilobytes -- before round 5 each stream had the large one
        // (~0.5 GB of an encoder's 6 GB for nothing).
        tmp_bytes_main_ = (s1 > s2 ? s1 : s2) + 256;
        tmp_bytes_side_ = s2 + 256;
        ORZ_HIP_CHECK(hipMalloc(&tmps_[0], tmp_bytes_main_));
        for (int i = 1; i < 3; i++) ORZ_HIP_CHECK(hipMalloc(&tmps_[i], tmp_bytes_side_));
        tmp_ = tmps_[0];
        tmp_bytes_ = tmp_bytes_main_;
    }
    ~HipBackend() {
        (void)hipSetDevice(device_);
        for (int i = 0; i < kStreams; i++) if (streams_[i]) (void)hipStreamSynchronize(streams_[i]);
        for (int i = 0; i < kStreams; i++) (void)hipFree(tmps_[i]);
        if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
        if (arena_) (void)hipFree(arena_);
        for (int i = 0; i < kEvents; i++) (void)hipEventDestroy(sev_[i]);
        for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
        for (hipEvent_t e : nev_) (void)hipEventDestroy(e);
        for (auto& kv : graph_exec_) (void)hipGraphExecDestroy(kv.second);
        StreamPool::Set t;  // (synchronised above: the streams go back to the pool idle)
        for (int i = 0; i < kStreams; i++) t.s[i] = streams_[i];
        if (streams_[0]) StreamPool::get().give(device_, rank_prio_, t);
    }
    HipBackend(const HipBackend&) = delete;
    HipBackend& operator=(const HipBackend&) = delete;

    hipStream_t stream() const { return streams_[0]; }
    // hand-off inside a sweep: polls a wave spends waiting for its predecessor's exit stamp before giving up
    uint32_t handoff_polls() const { return 1000; }
    uint32_t handoff_deadline() const { return 11000; }  // 110 us in ticks of the 100 MHz wall clock
    // far-from-the-front limits: measured counter-productive on MI355X (every evaluation of a far segment
    // pre-converges state the front later flies through), so off by default; ORZ_NEAR / ORZ_FAR_DEADLINE_US /
    // ORZ_SKIP_US turn them on for experiments
    uint32_t near_blocks() const { return 0; }
    uint32_t far_deadline() const { return 0; }
    uint32_t skip_after() const { return 0; }
    // streams 1 and 2: the tail stage of a block (symbol ranking; Huffman + packing) overlaps the next block's parse (orz_stream.h)
    void select(int s) { stream_ = streams_[s]; tmp_ = tmps_[s]; tmp_bytes_ = s == 0 ? tmp_bytes_main_ : (s < 3 ? tmp_bytes_side_ : 0); cur_ = s; }
    void record(int ev) { ORZ_HIP_CHECK(hipEventRecord(sev_[ev], stream_)); }
    void wait(int ev) { ORZ_HIP_CHECK(hipStreamWaitEvent(stream_, sev_[ev], 0)); }
    int device() const { return device_; }
    // wave slots the parse kernel can fill at once: its LDS per wave against 160 KB per CU, at most 3 per SIMD
    // (142 VGPRs).  The sweep window is sized to this: waves of one launch hand their exits on to each other,
    // so a launch that does not fit the chip runs in several rounds and takes that many times longer.
    uint32_t resident_parse_waves(size_t lds_bytes) const {
        int cus = 0;
        ORZ_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_));
        const size_t gran = 1280;  // LDS allocation granule
        const size_t per_wave = (lds_bytes + gran - 1) / gran * gran;
        size_t per_cu = per_wave ? (160 * 1024) / per_wave : 12;
        if (per_cu > 12) per_cu = 12;
        if (per_cu < 1) per_cu = 1;
        return (uint32_t)(per_cu * (size_t)(cus > 0 ? cus : 256));
    }

    // ORZ_ARENA_MB=<n> (diagnostics, off by default): an encoder's ~60 buffers are carved out of ONE device allocation, 2 MiB-
    // aligned when large, 256 B otherwise; nothing is returned before the last of them is freed, a buffer that does not fit
    // falls back to hipMalloc.  Built in round 4 to test whether allocation placement explains why a process's later
    // encoders are slower (it does not: the streams' hardware queues do, see StreamPool).  What it is good for: buffers that
    // lie side by side turn an out-of-bounds access into a visible defect -- with it the exact mode's object-level test
    // tripped the validity gate: rebuild_summaries (orz_parse.h) wrote 62 words past the word list's level-1 summary on
    // every full block, into allocator slack with separate allocations, into the level-2 summary with neighbours.  Fixed;
    // the knob stays for the next defect of that kind (the emulation has the same: ORZ_EMU_ARENA_MB).
    template <class T>
    T* alloc(size_t n, bool zero = true) {
        const size_t bytes = (n ? n : 1) * sizeof(T);
        void* p = arena_take(bytes);
        if (!p) {
            ORZ_HIP_CHECK(hipSetDevice(device_));
            ORZ_HIP_CHECK(hipMalloc(&p, bytes));
        }
        if (zero) ORZ_HIP_CHECK(hipMemsetAsync(p, 0, bytes, stream_));
        return (T*)p;
    }
    void free(void* p) {
        if (p && arena_ && (char*)p >= arena_ && (char*)p < arena_ + arena_bytes_) {
            // (space comes back when the LAST buffer carved from the arena is gone: an encoder's buffers die together, and a
            // rebuilt encoder may share the arena with its predecessor for a moment)
            if (--arena_live_ == 0) {
                for (int i = 0; i < kStreams; i++) if (streams_[i]) (void)hipStreamSynchronize(streams_[i]);
                arena_used_ = 0;
            }
            return;
        }
        (void)hipFree(p);
    }
    void release_arena() {}
    void enable_arena() { arena_on_ = true; }  // (stream encoders only: the decoder's and the Huffman entry point's backends allocate as before)
    void poison(void*, size_t) {}  // (the emulation backend fills state that must never be read before it is written)
    void memset(void* p, int v, size_t n) {
        if (!n) return;
        Bracket br(*this, profile_ ? KernelNames::lib("(fill)") : 0);
        ORZ_HIP_CHECK(hipMemsetAsync(p, v, n, stream_));
    }
    void h2d(void* d, const void* s, size_t n) {
        if (!n) return;
        ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream_));
        ORZ_HIP_CHECK(hipStreamSynchronize(stream_));  // pageable source may be reused by the caller
        nsync_++;
    }
    // host -> device from PINNED host memory (hipHostMalloc / hipHostRegister): asynchronous on the encoder's stream, no
    // synchronisation here -- the caller must not reuse the source before the stream's next sync (encode_block ends in one)
    void h2d_pinned(void* d, const void* s, size_t n) {
        if (!n) return;
        ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream_));
    }
    void d2h(void* d, const void* s, size_t n) {
        if (!n) return;
        ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream_));
        ORZ_HIP_CHECK(hipStreamSynchronize(stream_));
        nsync_++;
    }
    // device -> host without waiting: the caller reads `d` after the stream's next sync()
    void d2h_async(void* d, const void* s, size_t n) {
        if (n) ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream_));
    }
    // times the host waited for a stream since the last call (orz_encode_stats.host_syncs): every hipStreamSynchronize
    // the encoder's own code makes -- reads of counts and sizes, the copies of finished output
    uint64_t take_host_syncs() { const uint64_t v = nsync_; nsync_ = 0; return v; }
    void d2d(void* d, const void* s, size_t n) {
        if (!n) return;
        const char* a = (const char*)s;
        char* b = (char*)d;
        if ((b < a + n && a < b + n)) {  // overlapping (window slide): go through chunks in address order
            if (b > a) throw std::runtime_error("d2d: forward-overlapping copy unsupported");
            const size_t gap = (size_t)(a - b);
            for (size_t off = 0; off < n; off += gap) {
                size_t m = n - off < gap ? n - off : gap;
                ORZ_HIP_CHECK(hipMemcpyAsync(b + off, a + off, m, hipMemcpyDeviceToDevice, stream_));
            }
        } else {
            ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream_));
        }
    }
    void sync() { ORZ_HIP_CHECK(hipStreamSynchronize(stream_)); nsync_++; }
    // the closing time stamp of an encode whose stream stays on the device: recorded on the stream that finishes it (the copy
    // stream, behind the last frame kernel), so that no extra wait on the main stream is needed to read the elapsed time
    void set_end_event(hipEvent_t e) { end_ev_ = e; }
    void mark_end() { if (end_ev_) ORZ_HIP_CHECK(hipEventRecord(end_ev_, stream_)); }
    void parse_token_acquire() { if (!lone_) ParseTokens::of(device_).acquire(); }
    void parse_token_release() { if (!lone_) ParseTokens::of(device_).release(); }
    // counts the host derives instead of reading them back are verified against the device only on request
    bool check_hints() const { static const bool on = getenv("ORZ_CHECK_HINTS") != nullptr; return on; }
    double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    // profile mode: every launch between two events on its stream, summed per kernel name by collect_named()
    struct Bracket {
        HipBackend& be;
        bool on;
        Bracket(HipBackend& b, int id) : be(b), on(b.profile_ && !b.capturing_) {
            if (on) be.named_begin(id);
        }
        ~Bracket() {
            if (on) be.named_end();
        }
    };
    template <class F>
    void launch(size_t n, const F& f) {
        if (!n) return;
        Bracket br(*this, profile_ ? KernelNames::of<F>() : 0);
        const unsigned grid = (unsigned)((n + 255) / 256);
#if defined(ORZ_EVAL_WAVES)
        if constexpr (std::is_same<F, FastEval>::value) {
            hipLaunchKernelGGL(orz_thread_kernel_occ<F>, dim3(grid), dim3(256), 0, stream_, f, n);
            ORZ_HIP_CHECK(hipGetLastError());
            return;
        }
#endif
        hipLaunchKernelGGL(orz_thread_kernel<F>, dim3(grid), dim3(256), 0, stream_, f, n);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    template <class K>
    void launch_waves(size_t nblocks, const K& k, size_t lds_bytes) {
        if (!nblocks) return;
        Bracket br(*this, profile_ ? KernelNames::of<K>() : 0);
        if (lds_bytes > 64 * 1024) {  // more than 64 KB of dynamic LDS (a CU of gfx950 has 160 KB) has to be asked for once per kernel
            static const hipError_t big = hipFuncSetAttribute(reinterpret_cast<const void*>(&orz_wave_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            if (big != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(big));
        }
        hipLaunchKernelGGL(orz_wave_kernel<K>, dim3((unsigned)nblocks), dim3(64), lds_bytes, stream_, k);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    template <class K>
    void launch_group(const K& k) {
        Bracket br(*this, profile_ ? KernelNames::of<K>() : 0);
        size_t lds = k.lds_bytes();
        // more than 64 KB of dynamic LDS (a CU of gfx950 has 160 KB) has to be asked for once per kernel
        static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&orz_group_kernel<K>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::kLdsMax) == hipSuccess;
        if (lds > 64 * 1024 && !big_ok) lds = 0;  // (the kernel then works on global memory)
        hipLaunchKernelGGL(orz_group_kernel<K>, dim3(1), dim3(1024), lds, stream_, k, lds != 0);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    void huffbuild(const HuffBuild& f) {
        if (f.nchunks) launch_waves((size_t)f.nchunks * 3, HuffWave{f}, HuffWave::lds_bytes());
    }
    void rank(const RankArgs& a, uint32_t nchunks) {
        const uint32_t nvblk = (a.nvwords + 4095) / 4096, nkblk = (a.nkwords + 4095) / 4096;
        hipLaunchKernelGGL(orz_rank_kernel, dim3(nchunks + nvblk + nkblk), dim3(256), 0, stream_, a, nchunks, nvblk);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    void sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits) {
        if (n == 0) return;
        Bracket br(*this, profile_ ? KernelNames::lib("(radix sort, pairs)") : 0);
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::radix_sort_pairs(tmp_, sz, kin, kout, vin, vout, n, 0, (unsigned)bits, stream_));
    }
    // stable sort of the item indices 0..n-1 by their 9-bit symbol-ranking context
    void sort_by_ctx(const uint16_t* ctx, uint16_t* ctx_sorted, uint32_t* perm, size_t n) {
        if (n == 0) return;
        Bracket br(*this, profile_ ? KernelNames::lib("(radix sort, items by context)") : 0);
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::radix_sort_pairs(tmp_, sz, ctx, ctx_sorted, rocprim::counting_iterator<uint32_t>(0), perm, n, 0, 9, stream_));
    }
    // stable sort of 64-bit keys on their bits [begin_bit, bits): the callers' keys come in the order of their low part (item
    // index / position), so only the high part has to be sorted -- half the passes (round 5)
    const uint64_t* sort_u64(uint64_t* a, uint64_t* b, size_t n, int bits, int begin_bit = 0) {
        if (n == 0) return a;
        Bracket br(*this, profile_ ? KernelNames::lib("(radix sort, 64-bit keys)") : 0);
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::radix_sort_keys(tmp_, sz, a, b, n, (unsigned)begin_bit, (unsigned)bits, stream_));
        return b;
    }
    void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        if (n == 0) return;
        Bracket br(*this, profile_ ? KernelNames::lib("(scan)") : 0);
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::exclusive_scan(tmp_, sz, in, out, 0u, n, rocprim::plus<uint32_t>(), stream_));
    }
    void inclusive_max_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        if (n == 0) return;
        Bracket br(*this, profile_ ? KernelNames::lib("(scan, running maximum)") : 0);
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::inclusive_scan(tmp_, sz, in, out, n, rocprim::maximum<uint32_t>(), stream_));
    }
    // HIP-event brackets around selected kernels (bench.py roofline leg): slot 0 = the parse's per-position kernel
    // (ParseWave / FastEval), 1 = symbol ranking, 2 = candidate table build, 3 = path extraction.  Events are recorded on
    // the stream the kernel is launched on.
    static constexpr int kTimedSlots = 4;
    void timed_begin(int slot = 0) {
        if (!timing_ || capturing_ || ((slot == 0 || slot == 3) && !profile_)) return;
        if (ev_used_ + 2 > ev_.size()) {
            for (int i = 0; i < 256; i++) {
                hipEvent_t e;
                ORZ_HIP_CHECK(hipEventCreate(&e));
                ev_.push_back(e);
            }
        }
        ORZ_HIP_CHECK(hipEventRecord(ev_[ev_used_], stream_));
        ev_slot_.resize(ev_.size() / 2 + 1);
        ev_slot_[ev_used_ / 2] = slot;
    }
    void timed_end(int slot = 0) {
        if (!timing_ || capturing_ || ((slot == 0 || slot == 3) && !profile_)) return;
        ORZ_HIP_CHECK(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
    }
    void set_timing(bool on) { timing_ = on; }
    void named_begin(int id) {
        if (nev_used_ + 2 > nev_.size()) {
            for (int i = 0; i < 1024; i++) {
                hipEvent_t e;
                ORZ_HIP_CHECK(hipEventCreate(&e));
                nev_.push_back(e);
            }
            nev_id_.resize(nev_.size() / 2);
        }
        if (nest_++) return;  // (a library call inside a bracketed one counts once)
        ORZ_HIP_CHECK(hipEventRecord(nev_[nev_used_], stream_));
        nev_id_[nev_used_ / 2] = id;
    }
    void named_end() {
        if (--nest_) return;
        ORZ_HIP_CHECK(hipEventRecord(nev_[nev_used_ + 1], stream_));
        nev_used_ += 2;
    }
    // per kernel name: summed event time and launch count since the last call (all streams synchronised first)
    std::vector<KernelRow> collect_named() {
        for (int i = 0; i < kStreams; i++) ORZ_HIP_CHECK(hipStreamSynchronize(streams_[i]));
        std::map<int, KernelRow> acc;
        for (size_t i = 0; i + 1 < nev_used_; i += 2) {
            float t = 0;
            ORZ_HIP_CHECK(hipEventElapsedTime(&t, nev_[i], nev_[i + 1]));
            KernelRow& r = acc[nev_id_[i / 2]];
            r.ms += t;
            r.launches++;
        }
        nev_used_ = 0;
        std::vector<KernelRow> out;
        for (auto& kv : acc) {
            kv.second.name = KernelNames::get().name(kv.first);
            out.push_back(kv.second);
        }
        std::sort(out.begin(), out.end(), [](const KernelRow& a, const KernelRow& b) { return a.ms > b.ms; });
        return out;
    }
    // sums of the bracketed intervals in ms per slot since the last call, and their counts; returns slot 0's sum
    double collect_timed(uint64_t* launches, double* ms_by_slot = nullptr, uint64_t* n_by_slot = nullptr) {
        for (int i = 0; i < kStreams; i++) ORZ_HIP_CHECK(hipStreamSynchronize(streams_[i]));
        double ms[kTimedSlots] = {0, 0, 0, 0};
        uint64_t cnt[kTimedSlots] = {0, 0, 0, 0};
        for (size_t i = 0; i + 1 < ev_used_; i += 2) {
            float t = 0;
            ORZ_HIP_CHECK(hipEventElapsedTime(&t, ev_[i], ev_[i + 1]));
            const int sl = ev_slot_[i / 2] & (kTimedSlots - 1);
            ms[sl] += t;
            cnt[sl]++;
        }
        if (launches) *launches = cnt[0];
        for (int i = 0; i < kTimedSlots; i++) {
            if (ms_by_slot) ms_by_slot[i] = ms[i];
            if (n_by_slot) n_by_slot[i] = cnt[i];
        }
        ev_used_ = 0;
        return ms[0];
    }
    // hipGraph of a launch sequence that is identical from block to block (the fast parse's round loop over a full
    // block: ~2,400 small kernels): captured once on the encoder's stream, replayed with one call per block.
    bool graphs_enabled() const { return graphs_ && !profile_; }
    bool graph_replay(uint64_t key) {
        auto it = graph_exec_.find(key);
        if (it == graph_exec_.end()) return false;
        ORZ_HIP_CHECK(hipGraphLaunch(it->second, stream_));
        return true;
    }
    // (captured on a stream of the backend's own; nothing executes during a capture, the graph is launched on the main stream)
    void graph_capture_begin() {
        if (!cap_stream_) ORZ_HIP_CHECK(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
        run_stream_ = stream_;
        stream_ = cap_stream_;
        ORZ_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        capturing_ = true;
    }
    void graph_capture_end(uint64_t key) {
        hipGraph_t g = nullptr;
        capturing_ = false;
        const hipError_t ce = hipStreamEndCapture(stream_, &g);
        stream_ = run_stream_;
        ORZ_HIP_CHECK(ce);
        hipGraphExec_t ex = nullptr;
        hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) throw std::runtime_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        graph_exec_[key] = ex;
        ORZ_HIP_CHECK(hipGraphLaunch(ex, stream_));
    }
    // a launch threw inside the capture: end it, drop the partial graph, leave the stream usable
    void graph_capture_abort() {
        if (!capturing_) return;
        capturing_ = false;
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(stream_, &g);
        stream_ = run_stream_;
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
    }
    void set_graphs(bool on) { graphs_ = on; }
    // the captured launches hold the encoder's buffer addresses and settings: forget them when the encoder is rebuilt
    void clear_graphs() {
        for (auto& kv : graph_exec_) (void)hipGraphExecDestroy(kv.second);
        graph_exec_.clear();
    }
    // profile mode: HIP-event brackets around the kernels inside the round loop (so no graph replay)
    void set_profile(bool on) { profile_ = on; nev_used_ = 0; nest_ = 0; }
    // the brackets of a profiled encode nobody collected (stats == NULL) must not pile up, nor leak into the next table
    void begin_encode() { nev_used_ = 0; nest_ = 0; ev_used_ = 0; }
    bool profile() const { return profile_; }

    // Symbol ranking of one block, guarded.  The ranks of the hand-scheduled kernel are checked against the one property
    // that needs no table -- rank 388 exactly where the symbol is the excluded one (SymCheck) -- and when the check fails
    // the block is ranked again from the tables saved before the first run; what the second run left is checked again into
    // flags[1], which the host reads with the block's output (a set flags[1] fails the encode: no stream is better than a
    // wrong one).  An invariant check: it was written when invalid members were blamed on this kernel; their cause
    // turned out to be elsewhere, and in soaks with every block ranked twice the two runs never differed.
    // ORZ_SYMRANK_VERIFY=1 (diagnostics): every block is ranked twice from the same tables and the two runs' ranks are
    // compared (flags[2] = differences; `keep` holds the first run's ranks) -- twice the chain, for soak runs only.
    void symrank(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank, const uint32_t* rstart, uint32_t nitems, uint32_t* flags,
                 uint16_t* backup, uint16_t* keep) {
        static_assert((512 * kSrWords * 2) % 8 == 0, "the tables are copied in 8-byte words");
        const uint32_t nw = 512 * kSrWords * 2 / 8;
        launch(nw, SymGuardBegin{reinterpret_cast<const uint64_t*>(srstate), reinterpret_cast<uint64_t*>(backup), nw, flags});
        Bracket br(*this, profile_ ? KernelNames::lib("orz_symrank_kernel (+ guard)") : 0);
        timed_begin(1);
        hipLaunchKernelGGL(orz_symrank_kernel, dim3(512), dim3(64), 0, stream_, srstate, gsym, grank, rstart, (const uint16_t*)nullptr,
                           (const uint32_t*)nullptr);
        ORZ_HIP_CHECK(hipGetLastError());
        timed_end(1);
        const char* inj = getenv("ORZ_SYMRANK_INJECT");  // (tests: a wrong rank at item k of every block)
        const long inject = inj ? atol(inj) : -1;
        if (inject >= 0 && (uint32_t)inject < nitems) launch(1, SymInject{gsym, grank, (uint32_t)inject});
        launch(SymCheck::kThreads, SymCheck{gsym, grank, nitems, flags, nullptr});
        hipLaunchKernelGGL(orz_symrank_kernel, dim3(512), dim3(64), 0, stream_, srstate, gsym, grank, rstart, (const uint16_t*)backup,
                           (const uint32_t*)flags);
        ORZ_HIP_CHECK(hipGetLastError());
        launch(SymCheck::kThreads, SymCheck{gsym, grank, nitems, flags + 1, flags});
        static const bool verify = getenv("ORZ_SYMRANK_VERIFY") && atoi(getenv("ORZ_SYMRANK_VERIFY"));
        if (verify && keep) {
            launch(SymCompare::kThreads, SymKeep{grank, keep, nitems});
            hipLaunchKernelGGL(orz_symrank_kernel, dim3(512), dim3(64), 0, stream_, srstate, gsym, grank, rstart, (const uint16_t*)backup,
                               (const uint32_t*)nullptr);
            ORZ_HIP_CHECK(hipGetLastError());
            launch(SymCompare::kThreads, SymCompare{grank, keep, nitems, flags + 2});
        }
    }

   private:
    void* arena_take(size_t bytes) {
        if (!arena_on_) return nullptr;
        if (!arena_tried_) {
            arena_tried_ = true;
            const char* v = getenv("ORZ_ARENA_MB");
            const size_t mb = v ? (size_t)strtoull(v, nullptr, 10) : 0;
            if (mb) {
                (void)hipSetDevice(device_);
                void* p = nullptr;
                if (hipMalloc(&p, mb << 20) == hipSuccess) { arena_ = (char*)p; arena_bytes_ = mb << 20; }
                else (void)hipGetLastError();
            }
        }
        if (!arena_) return nullptr;
        const size_t align = bytes >= (1u << 20) ? (2u << 20) : 256;
        const size_t at = (arena_used_ + align - 1) / align * align;
        if (at + bytes > arena_bytes_) return nullptr;
        arena_used_ = at + bytes;
        arena_live_++;
        return arena_ + at;
    }
    char* arena_ = nullptr;
    size_t arena_bytes_ = 0, arena_used_ = 0, arena_live_ = 0;
    bool arena_tried_ = false, arena_on_ = false;
    int device_;
    bool lone_ = true, rank_prio_ = false;
    hipStream_t stream_ = nullptr;
    hipStream_t cap_stream_ = nullptr, run_stream_ = nullptr;  // graph capture (see graph_capture_begin)
    hipEvent_t end_ev_ = nullptr;
    static constexpr int kStreams = StreamPool::kStreams, kEvents = 16;  // (stream 3 only copies finished output to the host: no temporary storage)
    hipStream_t streams_[kStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sev_[kEvents];
    void* tmps_[kStreams] = {nullptr, nullptr, nullptr, nullptr};
    int cur_ = 0;
    void* tmp_ = nullptr;
    size_t tmp_bytes_ = 0, tmp_bytes_main_ = 0, tmp_bytes_side_ = 0;
    uint64_t nsync_ = 0;
    bool timing_ = false;
    std::vector<hipEvent_t> ev_;
    std::vector<int> ev_slot_;
    std::vector<hipEvent_t> nev_;  // profile mode: event pairs around every launch, with the kernel's name id
    std::vector<int> nev_id_;
    size_t nev_used_ = 0;
    int nest_ = 0;
    std::map<uint64_t, hipGraphExec_t> graph_exec_;
    bool graphs_ = true, profile_ = false, capturing_ = false;
    size_t ev_used_ = 0;
};

}  // namespace orz
