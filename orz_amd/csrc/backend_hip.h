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

// SymRankCoder chains (src/symrank.rs:38-97): one wavefront per context.  The chain is serial by definition and a
// lone wavefront issues one instruction every four cycles, so what counts is the number of instructions per item.
// The 64 best-ranked symbols live in ONE vector register (lane r holds value[r]): "which rank has symbol v" is a
// compare + s_ff1, the 3-way rotate is two v_readlane and three v_writelane with scalar operands, and the whole
// loop runs on the scalar unit -- no LDS round trip on the common path.  Ranks >= 64 stay in LDS (value[] for
// those ranks; index[] for the symbols that sit there).  Items are fetched 64 at a time into a register and
// handed out with v_readlane; the ranks go back through v_writelane the same way.
__device__ __forceinline__ int orz_writelane(int old, uint32_t sval, uint32_t slane) {  // old[slane] = sval (both uniform)
    // (VOP3 reads one SGPR only: the lane select travels in M0 -- which the compiler does not let an asm statement clobber
    // ("may not be preserved across the asm statement"), so the statement puts back what it found there)
    uint32_t m0_was;
    asm volatile("s_mov_b32 %[sv], m0\n\ts_mov_b32 m0, %[ln]\n\tv_writelane_b32 %[o], %[val], m0\n\ts_mov_b32 m0, %[sv]"
                 : [o] "+v"(old), [sv] "=&s"(m0_was)
                 : [val] "s"(sval), [ln] "s"(slane));
    return old;
}
__device__ __forceinline__ uint32_t orz_ff1(uint64_t m) {  // index of the lowest set bit, 0xffffffff for 0 (s_ff1_i32_b64)
    uint32_t r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t orz_sub_sat(uint32_t a, uint32_t b) {  // max(a - b, 0) on the scalar unit
    uint32_t r;
    asm("s_sub_u32 %0, %1, %2\n\ts_cselect_b32 %0, 0, %0" : "=&s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}
// `state_in` / `only_if`: the guarded second run of a block (HipBackend::symrank) -- it starts from the saved tables and
// does nothing unless the check of the first run's ranks raised *only_if.
__global__ __launch_bounds__(64) void orz_symrank_kernel(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank,
                                                         const uint32_t* rstart, const uint16_t* state_in, const uint32_t* only_if) {
    __shared__ uint16_t val[kSyms + 3];
    __shared__ uint16_t idx[kSyms + 3];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (only_if && *only_if == 0) return;
    const uint32_t a = rstart[c], e = rstart[c + 1];
    if (a >= e) return;
    __builtin_amdgcn_s_setprio(3);  // one serial chain per wave: issue ahead of the parse kernels' waves sharing the SIMD
    uint16_t* state = srstate + (size_t)c * kSrWords;
    const uint16_t* sin = state_in ? state_in + (size_t)c * kSrWords : state;
    for (uint32_t i = lane; i < kSyms; i += 64) { val[i] = sin[i]; idx[i] = sin[kSyms + i]; }
    __syncthreads();
    int v0 = val[lane], v1 = val[64 + lane], v2 = val[128 + lane];  // ranks 0..63, 64..127 and 128..191 live in three registers
    uint32_t cnt = __builtin_amdgcn_readfirstlane((int)(sin[2 * kSyms] | ((uint32_t)sin[2 * kSyms + 1] << 16)));
    uint32_t sum = __builtin_amdgcn_readfirstlane((int)(sin[2 * kSyms + 2] | ((uint32_t)sin[2 * kSyms + 3] << 16)));
    // reciprocals of the steady-state counts 327 + lane: floor(n / d) == mulhi(n, floor(2^32 / d) + 1) for n < 2^17
    const int mreg = (int)(0xffffffffu / (327 + lane) + 1);
    // value of rank r / store x at rank r, wherever that rank lives
    auto get = [&](uint32_t r) -> uint32_t {
        if (r < 64) return (uint32_t)__builtin_amdgcn_readlane(v0, (int)r);
        if (r < 128) return (uint32_t)__builtin_amdgcn_readlane(v1, (int)(r - 64));
        if (r < 192) return (uint32_t)__builtin_amdgcn_readlane(v2, (int)(r - 128));
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)val[r]);
    };
    auto put = [&](uint32_t r, uint32_t x) {
        if (r < 64) v0 = orz_writelane(v0, x, r);
        else if (r < 128) v1 = orz_writelane(v1, x, r - 64);
        else if (r < 192) v2 = orz_writelane(v2, x, r - 128);
        else { val[r] = (uint16_t)x; idx[x & 0xffff] = (uint16_t)r; }
    };
    // Per-lane move targets of ranks lane, 64 + lane, 128 + lane for the current quotient q = floor(avg rank / 16): the
    // serial chain then fetches next_i / ni1 with two v_readlane instead of nine dependent scalar ops.  q moves rarely;
    // the straight loop below leaves to the general code when it does (and for ranks >= 192 and the warm-up counts).
    int nxt0 = 0, n1t0 = 0, nxt1 = 0, n1t1 = 0, nxt2 = 0, n1t2 = 0;
    uint32_t qtab = 0xffffffffu;  // no quotient reaches this: the first item goes through the general code and builds the tables
    auto targets = [](uint32_t r, uint32_t q, int& nxt, int& n1t) {
        const uint32_t dec = (r >> 4) + q, half = r >> 1;
        uint32_t nx = r > dec ? r - dec : 0;
        nx = nx > half ? nx : half;
        nxt = (int)nx;
        n1t = (int)(nx + ((r - nx) >> 1));
    };
    auto rebuild = [&](uint32_t q) {
        qtab = q;
        targets(lane, q, nxt0, n1t0);
        targets(64 + lane, q, nxt1, n1t1);
        targets(128 + lane, q, nxt2, n1t2);
    };
    // a batch = up to 64 items, top-aligned in the lanes: item n of nthis sits in lane 64 - nthis + n, so that the loop
    // counter kb = n - nthis (mod 2^32) selects its lane with its low six bits and ends the loop with its carry
    auto load_batch = [&](uint32_t j0) -> int {
        if (j0 >= e) return 0;
        const uint32_t off = e - j0 < 64 ? 64 - (e - j0) : 0;
        return lane >= off ? (int)gsym[j0 + lane - off] : 0;
    };
    int items = load_batch(a);
    for (uint32_t j0 = a; j0 < e; j0 += 64) {
        const int items_next = load_batch(j0 + 64);  // in flight while this batch runs
        const uint32_t nthis = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e - j0 < 64 ? e - j0 : 64));
        int vi = 0, vu = 0;  // per lane: rank of the item's symbol / of its excluded symbol (0xffffffff = behind the symbol's)
        uint32_t kb = 0u - nthis;
        while (kb != 0) {
            {   // items whose symbol sits in the 192 register-resident ranks, with count >= 326 and an unchanged quotient:
                // src/symrank.rs:58-100 as straight lines (ranks 0..63: 33 instructions an item).  Wait states between a
                // VALU-written SGPR and its use as operand (2) / lane select (4) are covered by the instruction order (the
                // assembler adds none inside inline asm), and independent work sits between a VALU result and the scalar
                // instruction consuming it (~10 ns each otherwise).  Lane selects above 63 address lane (select & 63).
                // The quotient test without a multiply: q stays Q exactly while 16 Q cnt <= sum < 16 (Q + 1) cnt, i.e.
                // while qa = sum - 16 Q cnt stays below qw = 16 cnt as unsigned numbers; an item adds rank - 16 Q to qa, 16 to qw.
                uint32_t g, g2, i, j, t, x, y, pv, rv;
                uint64_t ma, mb;
                kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)kb);  // (uniform already; pins it to an SGPR for the asm operand)
                const uint32_t qc = qtab << 4;
                uint32_t qw = cnt << 4, qa = sum - qtab * qw;  // (qtab 0xffffffff before the first item: qa >= qw, the test fails)
// One of the rarer register combinations to its end: the next item's symbols, value[i] <- value[y] <- value[x] <- the
// item's symbol G (each rank in the register that holds it), then on to the next item (the other half of the loop body)
#define ORZ_LEAF(G, G2, NEXT, Ri, Ry, Rx)                                                                                \
    "v_readlane_b32 %[" G2 "], %[items], %[kb]\n\t"                                                                      \
    "v_readlane_b32 %[pv], %[" Ry "], %[y]\n\tv_readlane_b32 %[rv], %[" Rx "], %[x]\n\ts_mov_b32 m0, %[i]\n\t"                \
    "v_writelane_b32 %[" Ri "], %[pv], m0\n\ts_mov_b32 m0, %[y]\n\tv_writelane_b32 %[" Ry "], %[rv], m0\n\t"                 \
    "s_mov_b32 m0, %[x]\n\tv_writelane_b32 %[" Rx "], %[" G "], m0\n\t"                                                     \
    "s_cmp_lg_u32 %[kb], 0\n\ts_cbranch_scc1 " NEXT "01b\n\ts_branch 9f\n\t"
#define ORZ_QCHK(L)                                                                                                      \
    "s_add_u32 %[qa], %[qa], %[i]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\ts_cbranch_scc1 " L "\n\t"
// One item.  P prefixes its labels; (G, U) hold its symbol / excluded symbol, (G2, U2) receive the next item's while this
// one's results are in flight: a scalar instruction behind a VALU instruction that writes an SGPR waits ~5 issue slots,
// so SGPR-writing VALU instructions are clustered and the lane writes (no SGPR result) sit between them and their users.
#define ORZ_SR_ITEM(P, G, G2)                                                                                            \
    P "01:\n\t"                                                                                                          \
    "s_cmp_ge_u32 %[qw], 0x1860\n\t" /* count 390: rescale first (src/symrank.rs:63-66) */                               \
    "s_mov_b32 m0, %[kb]\n\t"                                                                                            \
    "v_cmp_eq_u32_sdwa %[ma], %[" G "], %[v0] src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                      \
    "v_cmp_eq_u32_sdwa %[mb], %[" G "], %[v0] src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                      \
    "s_cbranch_scc1 " P "30f\n\t"                                                                                        \
    "s_add_u32 %[qw], %[qw], 16\n\t"                                                                                     \
    "s_add_u32 %[kb], %[kb], 1\n\t"                                                                                      \
    "s_ff1_i32_b64 %[i], %[ma]\n\t"                                                                                      \
    "s_ff1_i32_b64 %[j], %[mb]\n\t"                                                                                      \
    "s_cmp_lt_i32 %[i], 0\n\t"                                                                                           \
    "s_cbranch_scc1 " P "04f\n\t"                                                                                        \
    /* ranks 0..63 */                                                                                                    \
    ORZ_QCHK(P "31f")                                                                                                    \
    "v_readlane_b32 %[y], %[n1t0], %[i]\n\t"                                                                             \
    "v_readlane_b32 %[x], %[nxt0], %[i]\n\t"                                                                             \
    "v_writelane_b32 %[vi], %[i], m0\n\t"                                                                                \
    "v_writelane_b32 %[vu], %[j], m0\n\t"                                                                                \
    "v_readlane_b32 %[" G2 "], %[items], %[kb]\n\t"                                                                      \
    "v_readlane_b32 %[pv], %[v0], %[y]\n\t"                                                                              \
    "v_readlane_b32 %[rv], %[v0], %[x]\n\t"                                                                              \
    "s_mov_b32 m0, %[i]\n\t"                                                                                             \
    "v_writelane_b32 %[v0], %[pv], m0\n\t"                                                                               \
    "s_mov_b32 m0, %[y]\n\t"                                                                                             \
    "v_writelane_b32 %[v0], %[rv], m0\n\t"                                                                               \
    "s_mov_b32 m0, %[x]\n\t"                                                                                             \
    "v_writelane_b32 %[v0], %[" G "], m0\n\t"                                                                            \
    P "08:\n\t"                                                                                                          \
    "s_cmp_lg_u32 %[kb], 0\n\t"
// The rarer paths of one item (placed behind both straight lines)
#define ORZ_SR_SIDE(P, G, G2, NEXT)                                                                                      \
    /* count 390: cnt and sum scale by 9/10 */                                                                           \
    P "30:\n\t"                                                                                                          \
    "s_lshr_b32 %[t], %[qw], 4\n\t"                                                                                      \
    "s_mul_i32 %[x], %[t], %[qc]\n\t"                                                                                    \
    "s_add_u32 %[x], %[x], %[qa]\n\t"                                                                                    \
    "s_mul_i32 %[t], %[t], 9\n\t"                                                                                        \
    "s_mul_hi_u32 %[t], %[t], 0xcccccccd\n\t"                                                                            \
    "s_lshr_b32 %[t], %[t], 3\n\t"                                                                                       \
    "s_mul_i32 %[x], %[x], 9\n\t"                                                                                        \
    "s_mul_hi_u32 %[x], %[x], 0xcccccccd\n\t"                                                                            \
    "s_lshr_b32 %[x], %[x], 3\n\t"                                                                                       \
    "s_lshl_b32 %[qw], %[t], 4\n\t"                                                                                      \
    "s_mul_i32 %[t], %[t], %[qc]\n\t"                                                                                    \
    "s_sub_u32 %[qa], %[x], %[t]\n\t"                                                                                    \
    "s_branch " P "01b\n\t"                                                                                              \
    /* the quotient moved, or the rank is not in a register: undo, the general code takes the item */                    \
    P "31:\n\t"                                                                                                          \
    "s_sub_u32 %[qa], %[qa], %[i]\n\t"                                                                                   \
    "s_add_u32 %[qa], %[qa], %[qc]\n\t"                                                                                  \
    P "32:\n\t"                                                                                                          \
    "s_sub_u32 %[qw], %[qw], 16\n\t"                                                                                     \
    "s_sub_u32 %[kb], %[kb], 1\n\t"                                                                                      \
    "s_branch 9f\n\t"                                                                                                    \
    /* ranks 64..127; the excluded symbol only matters when it ranks ahead, so its search stops with the symbol's register */ \
    P "04:\n\t"                                                                                                          \
    "v_cmp_eq_u32_sdwa %[ma], %[" G "], %[v1] src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                      \
    "v_cmp_eq_u32_sdwa %[mb], %[" G "], %[v1] src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                      \
    "s_nop 1\n\t"                                                                                                        \
    "s_ff1_i32_b64 %[i], %[ma]\n\t"                                                                                      \
    "s_ff1_i32_b64 %[t], %[mb]\n\t"                                                                                      \
    "s_cmp_lt_i32 %[i], 0\n\t"                                                                                           \
    "s_cbranch_scc1 " P "05f\n\t"                                                                                        \
    "s_add_u32 %[i], %[i], 64\n\t"                                                                                       \
    "s_cmp_lt_i32 %[j], 0\n\t"                                                                                           \
    "s_cbranch_scc0 " P "41f\n\t"                                                                                        \
    "s_cmp_lt_i32 %[t], 0\n\t"                                                                                           \
    "s_cbranch_scc1 " P "41f\n\t"                                                                                        \
    "s_add_u32 %[j], %[t], 64\n\t"                                                                                       \
    P "41:\n\t"                                                                                                          \
    "v_readlane_b32 %[y], %[n1t1], %[i]\n\t"                                                                             \
    "v_readlane_b32 %[x], %[nxt1], %[i]\n\t"                                                                             \
    "v_writelane_b32 %[vi], %[i], m0\n\t" /* (m0: the item's lane, set at the top; the general code overwrites the   */ \
    "v_writelane_b32 %[vu], %[j], m0\n\t" /*  two if the quotient test sends the item there)                          */ \
    ORZ_QCHK(P "31b")                                                                                                    \
    "s_cmp_lt_u32 %[x], 64\n\t"                                                                                          \
    "s_cbranch_scc1 " P "11f\n\t"                                                                                        \
    ORZ_LEAF(G, G2, NEXT, "v1", "v1", "v1")                                                                              \
    P "11:\n\t"                                                                                                          \
    "s_cmp_lt_u32 %[y], 64\n\t"                                                                                          \
    "s_cbranch_scc1 " P "12f\n\t"                                                                                        \
    ORZ_LEAF(G, G2, NEXT, "v1", "v1", "v0")                                                                              \
    P "12:\n\t"                                                                                                          \
    ORZ_LEAF(G, G2, NEXT, "v1", "v0", "v0")                                                                              \
    /* ranks 128..191 (beyond: the general code, nothing changed so far) */                                              \
    P "05:\n\t"                                                                                                          \
    "v_cmp_eq_u32_sdwa %[ma], %[" G "], %[v2] src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                      \
    "v_cmp_eq_u32_sdwa %[mb], %[" G "], %[v2] src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                      \
    "s_nop 1\n\t"                                                                                                        \
    "s_ff1_i32_b64 %[i], %[ma]\n\t"                                                                                      \
    "s_ff1_i32_b64 %[pv], %[mb]\n\t"                                                                                     \
    "s_cmp_lt_i32 %[i], 0\n\t"                                                                                           \
    "s_cbranch_scc1 " P "32b\n\t"                                                                                        \
    "s_add_u32 %[i], %[i], 0x80\n\t"                                                                                     \
    "s_cmp_lt_i32 %[j], 0\n\t"                                                                                           \
    "s_cbranch_scc0 " P "51f\n\t"                                                                                        \
    "s_cmp_lt_i32 %[t], 0\n\t"                                                                                           \
    "s_cbranch_scc1 " P "07f\n\t"                                                                                        \
    "s_add_u32 %[j], %[t], 64\n\t"                                                                                       \
    "s_branch " P "51f\n\t"                                                                                              \
    P "07:\n\t"                                                                                                          \
    "s_cmp_lt_i32 %[pv], 0\n\t"                                                                                          \
    "s_cbranch_scc1 " P "51f\n\t"                                                                                        \
    "s_add_u32 %[j], %[pv], 0x80\n\t"                                                                                    \
    P "51:\n\t"                                                                                                          \
    "v_readlane_b32 %[y], %[n1t2], %[i]\n\t"                                                                             \
    "v_readlane_b32 %[x], %[nxt2], %[i]\n\t"                                                                             \
    "v_writelane_b32 %[vi], %[i], m0\n\t"                                                                                \
    "v_writelane_b32 %[vu], %[j], m0\n\t"                                                                                \
    ORZ_QCHK(P "31b")                                                                                                    \
    "s_cmp_lt_u32 %[x], 0x80\n\t" /* (x >= i/2 >= 64) */                                                                 \
    "s_cbranch_scc1 " P "21f\n\t"                                                                                        \
    ORZ_LEAF(G, G2, NEXT, "v2", "v2", "v2")                                                                              \
    P "21:\n\t"                                                                                                          \
    "s_cmp_lt_u32 %[y], 0x80\n\t"                                                                                        \
    "s_cbranch_scc1 " P "22f\n\t"                                                                                        \
    ORZ_LEAF(G, G2, NEXT, "v2", "v2", "v1")                                                                              \
    P "22:\n\t"                                                                                                          \
    ORZ_LEAF(G, G2, NEXT, "v2", "v1", "v1")
                uint32_t m0_was;  // (the block moves lane selects through M0 and restores it at its one exit)
                asm volatile(
                    "s_mov_b32 %[m0s], m0\n\t"
                    "v_readlane_b32 %[g], %[items], %[kb]\n\t"
                    "s_nop 0\n\t"
                    ORZ_SR_ITEM("1", "g", "g2")
                    "s_cbranch_scc0 9f\n\t"
                    ORZ_SR_ITEM("2", "g2", "g")
                    "s_cbranch_scc0 9f\n\t"
                    ORZ_SR_ITEM("3", "g", "g2")
                    "s_cbranch_scc0 9f\n\t"
                    ORZ_SR_ITEM("4", "g2", "g")
                    "s_cbranch_scc1 101b\n\t"
                    "s_branch 9f\n\t"
                    ORZ_SR_SIDE("1", "g", "g2", "2")
                    ORZ_SR_SIDE("2", "g2", "g", "3")
                    ORZ_SR_SIDE("3", "g", "g2", "4")
                    ORZ_SR_SIDE("4", "g2", "g", "1")
                    "9:\n\t"
                    "s_mov_b32 m0, %[m0s]"
                    : [m0s] "=&s"(m0_was), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [vi] "+v"(vi), [vu] "+v"(vu), [kb] "+s"(kb), [qa] "+s"(qa),
                      [qw] "+s"(qw), [g] "=&s"(g), [g2] "=&s"(g2), [i] "=&s"(i), [j] "=&s"(j),
                      [t] "=&s"(t), [x] "=&s"(x), [y] "=&s"(y), [pv] "=&s"(pv), [rv] "=&s"(rv), [ma] "=&s"(ma), [mb] "=&s"(mb)
                    : [items] "v"(items), [n1t0] "v"(n1t0), [nxt0] "v"(nxt0), [n1t1] "v"(n1t1), [nxt1] "v"(nxt1),
                      [n1t2] "v"(n1t2), [nxt2] "v"(nxt2), [qc] "s"(qc)
                    : "scc");
#undef ORZ_SR_ITEM
#undef ORZ_SR_SIDE
#undef ORZ_LEAF
#undef ORZ_QCHK
                cnt = qw >> 4;
                sum = qa + qtab * qw;
            }
            if (kb == 0) break;
            // the general item
            const uint32_t k = kb & 63;
            const uint32_t gk = (uint32_t)__builtin_amdgcn_readlane(items, (int)k), v = gk & 0xffff, vun = gk >> 16;
            // (a register lane holds its symbol in the low half; the straight loop leaves the item's excluded symbol in the high half)
            const int w0 = v0 & 0xffff, w1 = v1 & 0xffff, w2 = v2 & 0xffff;
            uint32_t i = orz_ff1(__ballot(w0 == (int)v));
            uint32_t iu = orz_ff1(__ballot(w0 == (int)vun));  // 0xffffffff = not among the first 64: behind any of those
            const bool fast = (int32_t)i >= 0;
            if (!fast) {
                const uint32_t i1 = orz_ff1(__ballot(w1 == (int)v));
                if ((int32_t)i1 >= 0) i = 64 + i1;
                else {
                    const uint32_t i2 = orz_ff1(__ballot(w2 == (int)v));
                    i = (int32_t)i2 >= 0 ? 128 + i2 : (uint32_t)__builtin_amdgcn_readfirstlane((int)idx[v]);
                }
                if ((int32_t)iu < 0) {
                    const uint32_t u1 = orz_ff1(__ballot(w1 == (int)vun));
                    if ((int32_t)u1 >= 0) iu = 64 + u1;
                    else {
                        const uint32_t u2 = orz_ff1(__ballot(w2 == (int)vun));
                        iu = (int32_t)u2 >= 0 ? 128 + u2 : (uint32_t)__builtin_amdgcn_readfirstlane((int)idx[vun]);
                    }
                }
            }
            if (__builtin_expect(cnt > kSyms, 0)) {  // src/symrank.rs:63-66
                cnt = cnt * 9 / 10;
                sum = sum * 9 / 10;
            }
            cnt += 1;
            sum += i;
            const uint32_t n16 = sum >> 4;
            uint32_t q;
            if (__builtin_expect(cnt >= 327, 1)) q = __umulhi(n16, (uint32_t)__builtin_amdgcn_readlane(mreg, (int)(cnt - 327)));
            else q = (uint32_t)__builtin_amdgcn_readfirstlane((int)((n16 / cnt) & 0xffff));  // (the division runs on the vector unit)
            if (q != qtab) rebuild(q);
            uint32_t next_i = orz_sub_sat(i, (i >> 4) + q);
            const uint32_t half = i >> 1;
            next_i = next_i > half ? next_i : half;
            const uint32_t ni1 = next_i + ((i - next_i) >> 1);
            // value[i] <- value[ni1] <- value[next_i] <- v  (for a one-step move ni1 == next_i and this is the swap;
            // for no move all three coincide and nothing changes: src/symrank.rs:75-96)
            if (i != next_i) {
                const uint32_t nv1 = get(ni1), nv2 = get(next_i);
                put(i, nv1);
                if (ni1 != next_i) put(ni1, nv2);
                put(next_i, v);
            }
            vi = orz_writelane(vi, i, k);
            vu = orz_writelane(vu, iu, k);
            kb++;
        }
        // src/symrank.rs:98-100: the excluded symbol's rank is skipped; the symbol itself being the excluded one codes as the last rank
        const uint32_t ri = (uint32_t)vi, ru = (uint32_t)vu;
        const uint32_t outr = ri == ru ? kSyms - 1 : ri - (ri > ru);
        const uint32_t off = 64 - nthis;
        if (lane >= off) grank[j0 + lane - off] = (uint16_t)outr;
        items = items_next;
    }
    // tables back to HBM: the registers' 192 ranks first
    val[lane] = (uint16_t)v0;
    val[64 + lane] = (uint16_t)v1;
    val[128 + lane] = (uint16_t)v2;
    idx[v0 & 0xffff] = (uint16_t)lane;
    idx[v1 & 0xffff] = (uint16_t)(64 + lane);
    idx[v2 & 0xffff] = (uint16_t)(128 + lane);
    __syncthreads();
    for (uint32_t i = lane; i < kSyms; i += 64) { state[i] = val[i]; state[kSyms + i] = idx[i]; }
    if (lane == 0) {
        state[2 * kSyms] = (uint16_t)cnt;
        state[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
        state[2 * kSyms + 2] = (uint16_t)sum;
        state[2 * kSyms + 3] = (uint16_t)(sum >> 16);
    }
}

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
