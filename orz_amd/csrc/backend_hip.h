// backend_hip.h -- the product backend: runs the kernel bodies of orz_kernels.h as HIP kernels on
// gfx950 (MI355X).  One HipBackend = one device + one HIP stream; every launch, copy and memset of
// a stream encoder is ordered on that stream (so HIP events recorded on it bracket exactly the
// encoder's device work).
//
// Library primitives used as plumbing: rocPRIM device radix sort / exclusive scan (the candidate
// list build).  Everything on the orz path itself is a hand-written kernel body.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <rocprim/rocprim.hpp>
#include <stdexcept>
#include <string>
#include <vector>

#include "orz_kernels.h"
#include "orz_parse.h"

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

// Huffman code lengths + canonical codes (src/huffman.rs:27-141): one wavefront per (chunk, table); the
// weights and the heap / tree scratch live in LDS, lane 0 runs the (serial, tie-break exact) construction.
__global__ __launch_bounds__(64) void orz_huff_kernel(HuffBuild f) {
    __shared__ uint32_t sc[HuffBuild::kHuffScratch];
    __shared__ uint32_t w0[kSyms];
    const uint32_t tid = blockIdx.x, ch = tid / 3, t = tid % 3;
    const uint32_t n = HuffBuild::table_syms(t);
    const uint32_t* src = f.hw + (size_t)ch * kHwStride + HuffBuild::table_off(t);
    for (uint32_t i = threadIdx.x; i < n; i += 64) w0[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) f.build(tid, w0, sc);
}

// SymRankCoder chains (src/symrank.rs:38-97): one wavefront per context, the 389-entry value and
// index tables live in LDS; lane 0 walks the context's item run (the chain is serial by
// definition), all 64 lanes move the tables in and out of LDS.
__global__ __launch_bounds__(64) void orz_symrank_kernel(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank,
                                                         const uint32_t* rstart) {
    __shared__ uint16_t tab[2 * kSyms + 4];
    __shared__ uint32_t magic[392];  // reciprocals for the rank update's division by the (small) item count
    const uint32_t c = blockIdx.x;
    const uint32_t a = rstart[c], e = rstart[c + 1];
    if (a >= e) return;
    uint16_t* state = srstate + (size_t)c * kSrWords;
    for (uint32_t i = threadIdx.x; i < kSrWords; i += 64) tab[i] = state[i];
    for (uint32_t d = threadIdx.x; d < 392; d += 64) magic[d] = d >= 2 ? (uint32_t)(0x100000000ull / d) + 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint16_t* value = tab;
        uint16_t* index = tab + kSyms;
        uint32_t cnt = tab[2 * kSyms] | ((uint32_t)tab[2 * kSyms + 1] << 16);
        uint32_t sum = tab[2 * kSyms + 2] | ((uint32_t)tab[2 * kSyms + 3] << 16);
        uint32_t g = gsym[a];
        for (uint32_t j = a; j < e; j++) {
            uint32_t gn = j + 1 < e ? gsym[j + 1] : 0;  // prefetch the next item behind the table update
            grank[j] = symrank_encode(value, index, cnt, sum, (uint16_t)(g & 0xffff), (uint16_t)(g >> 16), magic);
            g = gn;
        }
        tab[2 * kSyms] = (uint16_t)cnt;
        tab[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
        tab[2 * kSyms + 2] = (uint16_t)sum;
        tab[2 * kSyms + 3] = (uint16_t)(sum >> 16);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kSrWords; i += 64) state[i] = tab[i];
}

class HipBackend {
   public:
    explicit HipBackend(int device) : device_(device) {
        ORZ_HIP_CHECK(hipSetDevice(device_));
        ORZ_HIP_CHECK(hipStreamCreateWithFlags(&streams_[0], hipStreamNonBlocking));
        ORZ_HIP_CHECK(hipStreamCreateWithFlags(&streams_[1], hipStreamNonBlocking));
        stream_ = streams_[0];
        for (int i = 0; i < 2; i++) ORZ_HIP_CHECK(hipEventCreateWithFlags(&sev_[i], hipEventDisableTiming));
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
        tmp_bytes_ = (s1 > s2 ? s1 : s2) + 256;
        ORZ_HIP_CHECK(hipMalloc(&tmps_[0], tmp_bytes_));
        ORZ_HIP_CHECK(hipMalloc(&tmps_[1], tmp_bytes_));
        tmp_ = tmps_[0];
    }
    ~HipBackend() {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(streams_[0]);
        (void)hipStreamSynchronize(streams_[1]);
        (void)hipFree(tmps_[0]);
        (void)hipFree(tmps_[1]);
        for (int i = 0; i < 2; i++) (void)hipEventDestroy(sev_[i]);
        for (hipEvent_t e : ev_) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(streams_[0]);
        (void)hipStreamDestroy(streams_[1]);
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
    // second stream: the tail stage of a block overlaps the next block's parse (orz_stream.h)
    void select(int s) { stream_ = streams_[s]; tmp_ = tmps_[s]; cur_ = s; }
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

    template <class T>
    T* alloc(size_t n) {
        void* p = nullptr;
        ORZ_HIP_CHECK(hipSetDevice(device_));
        ORZ_HIP_CHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
        ORZ_HIP_CHECK(hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(T), stream_));
        return (T*)p;
    }
    void free(void* p) { (void)hipFree(p); }
    void memset(void* p, int v, size_t n) {
        if (n) ORZ_HIP_CHECK(hipMemsetAsync(p, v, n, stream_));
    }
    void h2d(void* d, const void* s, size_t n) {
        if (!n) return;
        ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream_));
        ORZ_HIP_CHECK(hipStreamSynchronize(stream_));  // pageable source may be reused by the caller
    }
    void d2h(void* d, const void* s, size_t n) {
        if (!n) return;
        ORZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream_));
        ORZ_HIP_CHECK(hipStreamSynchronize(stream_));
    }
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
    void sync() { ORZ_HIP_CHECK(hipStreamSynchronize(stream_)); }
    double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    template <class F>
    void launch(size_t n, const F& f) {
        if (!n) return;
        const unsigned grid = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(orz_thread_kernel<F>, dim3(grid), dim3(256), 0, stream_, f, n);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    template <class K>
    void launch_waves(size_t nblocks, const K& k, size_t lds_bytes) {
        if (!nblocks) return;
        hipLaunchKernelGGL(orz_wave_kernel<K>, dim3((unsigned)nblocks), dim3(64), lds_bytes, stream_, k);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    void huffbuild(const HuffBuild& f) {
        if (!f.nchunks) return;
        hipLaunchKernelGGL(orz_huff_kernel, dim3(f.nchunks * 3), dim3(64), 0, stream_, f);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    void rank(const RankArgs& a, uint32_t nchunks) {
        const uint32_t nvblk = (a.nvwords + 4095) / 4096, nkblk = (a.nkwords + 4095) / 4096;
        hipLaunchKernelGGL(orz_rank_kernel, dim3(nchunks + nvblk + nkblk), dim3(256), 0, stream_, a, nchunks, nvblk);
        ORZ_HIP_CHECK(hipGetLastError());
    }
    void sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits) {
        if (n == 0) return;
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::radix_sort_pairs(tmp_, sz, kin, kout, vin, vout, n, 0, (unsigned)bits, stream_));
    }
    const uint64_t* sort_u64(uint64_t* a, uint64_t* b, size_t n, int bits) {
        if (n == 0) return a;
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::radix_sort_keys(tmp_, sz, a, b, n, 0, (unsigned)bits, stream_));
        return b;
    }
    void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        if (n == 0) return;
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::exclusive_scan(tmp_, sz, in, out, 0u, n, rocprim::plus<uint32_t>(), stream_));
    }
    void inclusive_max_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
        if (n == 0) return;
        size_t sz = tmp_bytes_;
        ORZ_HIP_CHECK(rocprim::inclusive_scan(tmp_, sz, in, out, n, rocprim::maximum<uint32_t>(), stream_));
    }
    // HIP-event bracket around the dominant kernel's launches (bench.py roofline leg)
    void timed_begin() {
        if (!timing_ || cur_ != 0) return;
        if (ev_used_ + 2 > ev_.size()) {
            for (int i = 0; i < 256; i++) {
                hipEvent_t e;
                ORZ_HIP_CHECK(hipEventCreate(&e));
                ev_.push_back(e);
            }
        }
        ORZ_HIP_CHECK(hipEventRecord(ev_[ev_used_], stream_));
    }
    void timed_end() {
        if (!timing_ || cur_ != 0) return;
        ORZ_HIP_CHECK(hipEventRecord(ev_[ev_used_ + 1], stream_));
        ev_used_ += 2;
    }
    void set_timing(bool on) { timing_ = on; }
    // sum of the bracketed intervals in ms since the last call; also returns their count
    double collect_timed(uint64_t* launches) {
        ORZ_HIP_CHECK(hipStreamSynchronize(streams_[0]));
        double ms = 0;
        for (size_t i = 0; i + 1 < ev_used_; i += 2) {
            float t = 0;
            ORZ_HIP_CHECK(hipEventElapsedTime(&t, ev_[i], ev_[i + 1]));
            ms += t;
        }
        if (launches) *launches = ev_used_ / 2;
        ev_used_ = 0;
        return ms;
    }
    void symrank(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank, const uint32_t* rstart) {
        hipLaunchKernelGGL(orz_symrank_kernel, dim3(512), dim3(64), 0, stream_, srstate, gsym, grank, rstart);
        ORZ_HIP_CHECK(hipGetLastError());
    }

   private:
    int device_;
    hipStream_t stream_ = nullptr;
    hipStream_t streams_[2] = {nullptr, nullptr};
    hipEvent_t sev_[2];
    void* tmps_[2] = {nullptr, nullptr};
    int cur_ = 0;
    void* tmp_ = nullptr;
    size_t tmp_bytes_ = 0;
    bool timing_ = false;
    std::vector<hipEvent_t> ev_;
    size_t ev_used_ = 0;
};

}  // namespace orz
