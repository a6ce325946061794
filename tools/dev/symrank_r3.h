// dev only: the round-3 symbol-ranking kernel (value[] in registers, compare + s_ff1 lookups, v_readlane / v_writelane moves),
// kept beside tools/dev/symrank_bench.hip so the lanes formulation (orz_amd/csrc/orz_symrank.h) can be timed against it.
#pragma once
#include "../../orz_amd/csrc/orz_kernels.h"
namespace orz {
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
__global__ __launch_bounds__(64) void orz_symrank_kernel_r3(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank,
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

}  // namespace orz
