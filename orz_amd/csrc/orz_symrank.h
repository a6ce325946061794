// Symbol ranking (SymRankCoder, src/symrank.rs:38-97) -- round 6's formulation of the chain: positions in lanes.
//
// One wavefront per context, as before: the chain of a context is serial by definition and a lone wavefront is issue-bound
// (tools/dev/issue_bench.hip, cycles at 2.4 GHz: scalar 4.5, vector 5.5 -- 4.0 with the upper half of EXEC off --, a
// compare into VCC 9 / 8, into an SGPR pair 11, v_readlane 8, ds_write_b16 9, ds_read_u16 6, an LDS round trip ~45, an
// untaken branch 2.5, a taken one ~23), so the cost of an item is the sum of its instructions' prices.  The round-3 kernel
// kept value[] (rank -> symbol) in registers, found a symbol's rank by compare + s_ff1, moved symbols with v_readlane /
// v_writelane through M0 and branched on which register held each rank: ~75 ns an item for ranks 0..63, more beyond,
// 112...117 ns an item in the hottest context of the text workload.
//
// Here the wavefront takes 32 items at a time and tracks WHERE THEIR SYMBOLS ARE instead of what sits where: lane 2k holds
// the current rank of item k's symbol, lane 2k+1 the rank of its excluded symbol.  An update of the table is a rotation of
// three ranks (i -> next_i, next_i -> ni1, ni1 -> i: src/symrank.rs:75-96; a swap or nothing when they coincide), and
// applying it to the 64 tracked ranks is three compares and three selects whatever the ranks are -- no register choice, no
// branch.  Item k's rank is then one v_readlane with a constant lane (the group's items are unrolled).  value[] lives in LDS
// and is only needed for the symbols nobody tracks: the two displaced values travel as one read and one write with lane
// addresses, the write one item late (the LDS round trip hides behind the next item's instructions); a rank that holds a
// tracked symbol may be stale in LDS during the group and is rewritten from the lanes when the group ends.  Every item leaves
// a snapshot of the tracked ranks in LDS (one ds_write); item k's output is read from snapshot k.  The group's start rebuilds
// index[] from value[] (seven reads and seven writes a lane) and looks the next 64 symbols up.
//
// Count, sum and the quotient sum / 16 / count are seven scalar instructions and two branches an item -- and the quotient
// moves once in ~2,000 items.  In the steady state a group therefore runs WITHOUT them (ORZ_SRL_ITEM_S: 22 instructions an
// item) on the assumption that the quotient stays, and the assumption is checked afterwards from the group's 32 ranks by a
// prefix sum, one item per lane, the scaling by 9/10 included (its place follows from the count).  A failed check (one
// group in ~60 on text) puts value[] back and runs the checked group (ORZ_SRL_ITEM: 31 instructions an item).
// (16 items a group on 32 lanes with the upper half of EXEC off, all compares through VCC, the write-back in mid-item: all
// measured slower -- DESIGN.md 6a, profiles/r06_symrank_bench.txt.)
//
// The first items of a context's life (count < 192: the quotient moves by more than one an item) and the last < 32 of a
// launch go through a plain loop over index[] / value[] in LDS.
#pragma once
#include "orz_kernels.h"

namespace orz {

#if defined(__HIPCC__)

constexpr uint32_t kSrGroup = 32;                     // items a group
constexpr uint32_t kSrPad = 448;                      // both tables padded to 7 x 64 entries (the pad ranks hold pad symbols: no guards)
constexpr uint32_t kSrValOff = 0;                     // LDS bytes: value[448] (u16; an entry is the LDS address of its symbol's index[] slot)
constexpr uint32_t kSrIdxOff = 896;                   //            index[448] (u16)
constexpr uint32_t kSrSnapOff = 1792;                 //            snapshots [32][64] (u16)
constexpr uint32_t kSrLdsBytes = kSrSnapOff + kSrGroup * 128;

// One item of a group.  P: label prefix (unique per item), J2: lane of the item's symbol (2 k), SNAP: byte offset of the
// item's snapshot, PREV: "1" if the item before it (same group) left two displaced values to write back.
// Register roles: x tracked ranks; si the item's rank; qw = 16 count, qa = sum - 16 q count, qc = 16 q (the quotient
// q = sum / 16 / count stays while 0 <= qa < qw: one unsigned compare); vi / nx / y the three ranks of the rotation
// (uniform); ai / ay / anx their LDS addresses; n1 / n2 the displaced values.
// Wait states the assembler does not insert inside an asm statement (measured against hipcc's own code for the same
// pattern): VALU-written SGPR or VCC -> VALU reading it: 2; VALU-written VGPR -> v_readlane: 1.  An SALU instruction
// reading an SGPR fresh from v_readlane stalls ~5 slots: `si` is first read on the scalar side eleven slots after.
// (An untaken branch straight behind its compare waits ~4 cycles for SCC: each sits a few instructions behind.  The two
// displaced values travel as ONE read and ONE write with lane addresses: lane 0 moves value[ni1] to value[i], the other
// lanes value[next_i] to value[ni1] -- read before either is written, as the reference's temporaries are.)
#define ORZ_SRL_WB_1 "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aw], %[n]\n\t"
#define ORZ_SRL_WB_0 "s_nop 0\n\t"
#define ORZ_SRL_ITEM(P, J2, SNAP, PREV)                                                                                  \
    "ds_write_b16 %[l2], %[x] offset:" SNAP "\n\t"                                                                       \
    "v_readlane_b32 %[si], %[x], " J2 "\n\t"                                                                             \
    "s_cmpk_ge_u32 %[qw], 0x1860\n\t" /* count 390: scale by 9/10 first (src/symrank.rs:63-66) */                        \
    ORZ_SRL_WB_##PREV                                                                                                    \
    "v_mov_b32 %[vi], %[si]\n\t"                                                                                         \
    "s_cbranch_scc1 " P "30f\n\t"                                                                                        \
    P "31:\n\t"                                                                                                          \
    "s_add_u32 %[qw], %[qw], 16\n\t"                                                                                     \
    "s_add_u32 %[qa], %[qa], %[si]\n\t"                                                                                  \
    "s_sub_u32 %[qa], %[qa], %[qc]\n\t"                                                                                  \
    "s_cmp_ge_u32 %[qa], %[qw]\n\t"                                                                                      \
    "v_lshrrev_b32 %[t], 4, %[vi]\n\t"                                                                                   \
    "v_lshrrev_b32 %[h], 1, %[vi]\n\t"                                                                                   \
    "v_sub_u32 %[t], %[vi], %[t]\n\t"                                                                                    \
    "s_cbranch_scc1 " P "40f\n\t"                                                                                        \
    P "41:\n\t"                                                                                                          \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\t"                                                                                 \
    "v_cmp_eq_u32_e64 %[m1], %[x], %[vi]\n\t"                                                                            \
    "v_max3_i32 %[nx], %[t], %[h], 0\n\t" /* next_i = max(i - min(i, i/16 + q), i/2) */                                  \
    "v_add_u32 %[y], %[vi], %[nx]\n\t"                                                                                   \
    "v_cmp_eq_u32_e64 %[m3], %[x], %[nx]\n\t"                                                                            \
    "v_lshrrev_b32 %[y], 1, %[y]\n\t" /* ni1 = next_i + (i - next_i) / 2 = (i + next_i) / 2 */                           \
    "v_cndmask_b32_e64 %[rr], %[nx], %[y], %[m01]\n\t" /* ranks read: lane 0 ni1, the others next_i */                   \
    "v_cmp_eq_u32_e32 vcc, %[x], %[y]\n\t"                                                                               \
    "v_cndmask_b32_e64 %[rw], %[y], %[vi], %[m01]\n\t" /* ranks written: lane 0 i, the others ni1 */                     \
    "v_lshl_add_u32 %[ar], %[rr], 1, %[base]\n\t"                                                                        \
    "v_cndmask_b32_e64 %[x], %[x], %[y], %[m3]\n\t" /* next_i -> ni1 */                                                  \
    "ds_read_u16 %[n], %[ar]\n\t"                                                                                        \
    "v_cndmask_b32_e32 %[x], %[x], %[vi], vcc\n\t" /* ni1 -> i (after the first: a swap has ni1 == next_i) */            \
    "v_lshl_add_u32 %[aw], %[rw], 1, %[base]\n\t"                                                                        \
    "v_cndmask_b32_e64 %[x], %[x], %[nx], %[m1]\n\t" /* i -> next_i */
// The same item without the scalar chain: a group run on the assumption that the quotient stays what it is (it moves once
// in ~2,000 items of the text workload's hottest context) and checked afterwards from the group's 32 ranks -- see the kernel.
#define ORZ_SRL_WBS_1 "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aw], %[n]\n\t"
#define ORZ_SRL_WBS_0 "s_nop 1\n\t"
#define ORZ_SRL_ITEM_S(J2, SNAP, PREV)                                                                                   \
    "ds_write_b16 %[l2], %[x] offset:" SNAP "\n\t"                                                                       \
    "v_readlane_b32 %[si], %[x], " J2 "\n\t"                                                                             \
    ORZ_SRL_WBS_##PREV                                                                                                   \
    "v_mov_b32 %[vi], %[si]\n\t"                                                                                         \
    "v_mad_i32_i24 %[t], %[vi], 15, %[c15]\n\t" /* i - i/16 - q = (15 i + 15 - 16 q) >> 4 (arithmetic): q is the group's */ \
    "v_lshrrev_b32 %[h], 1, %[vi]\n\t"                                                                                   \
    "v_cmp_eq_u32_e64 %[m1], %[x], %[vi]\n\t"                                                                            \
    "v_ashrrev_i32 %[t], 4, %[t]\n\t"                                                                                    \
    "v_max3_i32 %[nx], %[t], %[h], 0\n\t"                                                                                \
    "v_add_u32 %[y], %[vi], %[nx]\n\t"                                                                                   \
    "v_cmp_eq_u32_e64 %[m3], %[x], %[nx]\n\t"                                                                            \
    "v_lshrrev_b32 %[y], 1, %[y]\n\t"                                                                                    \
    "v_cndmask_b32_e64 %[rr], %[nx], %[y], %[m01]\n\t"                                                                   \
    "v_cmp_eq_u32_e32 vcc, %[x], %[y]\n\t"                                                                               \
    "v_cndmask_b32_e64 %[rw], %[y], %[vi], %[m01]\n\t"                                                                   \
    "v_lshl_add_u32 %[ar], %[rr], 1, %[base]\n\t"                                                                        \
    "v_cndmask_b32_e64 %[x], %[x], %[y], %[m3]\n\t"                                                                      \
    "ds_read_u16 %[n], %[ar]\n\t"                                                                                        \
    "v_cndmask_b32_e32 %[x], %[x], %[vi], vcc\n\t"                                                                       \
    "v_lshl_add_u32 %[aw], %[rw], 1, %[base]\n\t"                                                                        \
    "v_cndmask_b32_e64 %[x], %[x], %[nx], %[m1]\n\t"
// The rare paths of one item, placed behind the group's straight line
#define ORZ_SRL_SIDE(P)                                                                                                  \
    P "30:\n\t" /* count and sum scale by 9/10; qa follows */                                                            \
    "s_lshr_b32 %[s0], %[qw], 4\n\t"                                                                                     \
    "s_mul_i32 %[s1], %[s0], %[qc]\n\t"                                                                                  \
    "s_add_u32 %[s1], %[s1], %[qa]\n\t"                                                                                  \
    "s_mul_i32 %[s0], %[s0], 9\n\t"                                                                                      \
    "s_mul_hi_u32 %[s0], %[s0], 0xcccccccd\n\t"                                                                          \
    "s_lshr_b32 %[s0], %[s0], 3\n\t"                                                                                     \
    "s_mul_i32 %[s1], %[s1], 9\n\t"                                                                                      \
    "s_mul_hi_u32 %[s1], %[s1], 0xcccccccd\n\t"                                                                          \
    "s_lshr_b32 %[s1], %[s1], 3\n\t"                                                                                     \
    "s_lshl_b32 %[qw], %[s0], 4\n\t"                                                                                     \
    "s_mul_i32 %[s0], %[s0], %[qc]\n\t"                                                                                  \
    "s_sub_u32 %[qa], %[s1], %[s0]\n\t"                                                                                  \
    "s_branch " P "31b\n\t"                                                                                              \
    P "40:\n\t" /* the quotient moved: step it until 0 <= qa < qw again */                                               \
    "s_cmp_lt_i32 %[qa], 0\n\t"                                                                                          \
    "s_cbranch_scc1 " P "42f\n\t"                                                                                        \
    "s_sub_u32 %[qa], %[qa], %[qw]\n\t"                                                                                  \
    "s_add_u32 %[q], %[q], 1\n\t"                                                                                        \
    "s_add_u32 %[qc], %[qc], 16\n\t"                                                                                     \
    "s_branch " P "43f\n\t"                                                                                              \
    P "42:\n\t"                                                                                                          \
    "s_add_u32 %[qa], %[qa], %[qw]\n\t"                                                                                  \
    "s_sub_u32 %[q], %[q], 1\n\t"                                                                                        \
    "s_sub_u32 %[qc], %[qc], 16\n\t"                                                                                     \
    P "43:\n\t"                                                                                                          \
    "s_cmp_ge_u32 %[qa], %[qw]\n\t"                                                                                      \
    "s_cbranch_scc1 " P "40b\n\t"                                                                                        \
    "s_branch " P "41b\n\t"

#define ORZ_SRL_GROUP \
    ORZ_SRL_ITEM("100", "0", "0", 0) \
    ORZ_SRL_ITEM("101", "2", "128", 1) \
    ORZ_SRL_ITEM("102", "4", "256", 1) \
    ORZ_SRL_ITEM("103", "6", "384", 1) \
    ORZ_SRL_ITEM("104", "8", "512", 1) \
    ORZ_SRL_ITEM("105", "10", "640", 1) \
    ORZ_SRL_ITEM("106", "12", "768", 1) \
    ORZ_SRL_ITEM("107", "14", "896", 1) \
    ORZ_SRL_ITEM("108", "16", "1024", 1) \
    ORZ_SRL_ITEM("109", "18", "1152", 1) \
    ORZ_SRL_ITEM("110", "20", "1280", 1) \
    ORZ_SRL_ITEM("111", "22", "1408", 1) \
    ORZ_SRL_ITEM("112", "24", "1536", 1) \
    ORZ_SRL_ITEM("113", "26", "1664", 1) \
    ORZ_SRL_ITEM("114", "28", "1792", 1) \
    ORZ_SRL_ITEM("115", "30", "1920", 1) \
    ORZ_SRL_ITEM("116", "32", "2048", 1) \
    ORZ_SRL_ITEM("117", "34", "2176", 1) \
    ORZ_SRL_ITEM("118", "36", "2304", 1) \
    ORZ_SRL_ITEM("119", "38", "2432", 1) \
    ORZ_SRL_ITEM("120", "40", "2560", 1) \
    ORZ_SRL_ITEM("121", "42", "2688", 1) \
    ORZ_SRL_ITEM("122", "44", "2816", 1) \
    ORZ_SRL_ITEM("123", "46", "2944", 1) \
    ORZ_SRL_ITEM("124", "48", "3072", 1) \
    ORZ_SRL_ITEM("125", "50", "3200", 1) \
    ORZ_SRL_ITEM("126", "52", "3328", 1) \
    ORZ_SRL_ITEM("127", "54", "3456", 1) \
    ORZ_SRL_ITEM("128", "56", "3584", 1) \
    ORZ_SRL_ITEM("129", "58", "3712", 1) \
    ORZ_SRL_ITEM("130", "60", "3840", 1) \
    ORZ_SRL_ITEM("131", "62", "3968", 1) \
    "s_branch 9f\n\t" \
    ORZ_SRL_SIDE("100") \
    ORZ_SRL_SIDE("101") \
    ORZ_SRL_SIDE("102") \
    ORZ_SRL_SIDE("103") \
    ORZ_SRL_SIDE("104") \
    ORZ_SRL_SIDE("105") \
    ORZ_SRL_SIDE("106") \
    ORZ_SRL_SIDE("107") \
    ORZ_SRL_SIDE("108") \
    ORZ_SRL_SIDE("109") \
    ORZ_SRL_SIDE("110") \
    ORZ_SRL_SIDE("111") \
    ORZ_SRL_SIDE("112") \
    ORZ_SRL_SIDE("113") \
    ORZ_SRL_SIDE("114") \
    ORZ_SRL_SIDE("115") \
    ORZ_SRL_SIDE("116") \
    ORZ_SRL_SIDE("117") \
    ORZ_SRL_SIDE("118") \
    ORZ_SRL_SIDE("119") \
    ORZ_SRL_SIDE("120") \
    ORZ_SRL_SIDE("121") \
    ORZ_SRL_SIDE("122") \
    ORZ_SRL_SIDE("123") \
    ORZ_SRL_SIDE("124") \
    ORZ_SRL_SIDE("125") \
    ORZ_SRL_SIDE("126") \
    ORZ_SRL_SIDE("127") \
    ORZ_SRL_SIDE("128") \
    ORZ_SRL_SIDE("129") \
    ORZ_SRL_SIDE("130") \
    ORZ_SRL_SIDE("131") \
    "9:\n\t"

#define ORZ_SRL_GROUP_S \
    ORZ_SRL_ITEM_S("0", "0", 0) \
    ORZ_SRL_ITEM_S("2", "128", 1) \
    ORZ_SRL_ITEM_S("4", "256", 1) \
    ORZ_SRL_ITEM_S("6", "384", 1) \
    ORZ_SRL_ITEM_S("8", "512", 1) \
    ORZ_SRL_ITEM_S("10", "640", 1) \
    ORZ_SRL_ITEM_S("12", "768", 1) \
    ORZ_SRL_ITEM_S("14", "896", 1) \
    ORZ_SRL_ITEM_S("16", "1024", 1) \
    ORZ_SRL_ITEM_S("18", "1152", 1) \
    ORZ_SRL_ITEM_S("20", "1280", 1) \
    ORZ_SRL_ITEM_S("22", "1408", 1) \
    ORZ_SRL_ITEM_S("24", "1536", 1) \
    ORZ_SRL_ITEM_S("26", "1664", 1) \
    ORZ_SRL_ITEM_S("28", "1792", 1) \
    ORZ_SRL_ITEM_S("30", "1920", 1) \
    ORZ_SRL_ITEM_S("32", "2048", 1) \
    ORZ_SRL_ITEM_S("34", "2176", 1) \
    ORZ_SRL_ITEM_S("36", "2304", 1) \
    ORZ_SRL_ITEM_S("38", "2432", 1) \
    ORZ_SRL_ITEM_S("40", "2560", 1) \
    ORZ_SRL_ITEM_S("42", "2688", 1) \
    ORZ_SRL_ITEM_S("44", "2816", 1) \
    ORZ_SRL_ITEM_S("46", "2944", 1) \
    ORZ_SRL_ITEM_S("48", "3072", 1) \
    ORZ_SRL_ITEM_S("50", "3200", 1) \
    ORZ_SRL_ITEM_S("52", "3328", 1) \
    ORZ_SRL_ITEM_S("54", "3456", 1) \
    ORZ_SRL_ITEM_S("56", "3584", 1) \
    ORZ_SRL_ITEM_S("58", "3712", 1) \
    ORZ_SRL_ITEM_S("60", "3840", 1) \
    ORZ_SRL_ITEM_S("62", "3968", 1)

// `state_in` / `only_if`: the guarded second run of a block (HipBackend::symrank) -- it starts from the saved tables and
// does nothing unless the check of the first run's ranks raised *only_if.
__global__ __launch_bounds__(64) void orz_symrank_kernel(uint16_t* srstate, const uint32_t* gsym, uint16_t* grank,
                                                         const uint32_t* rstart, const uint16_t* state_in, const uint32_t* only_if) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[kSrLdsBytes / 2];
    uint16_t* const val = lds + kSrValOff / 2;
    uint16_t* const idx = lds + kSrIdxOff / 2;
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (only_if && *only_if == 0) return;
    const uint32_t a = rstart[c], e = rstart[c + 1];
    if (a >= e) return;
    __builtin_amdgcn_s_setprio(3);  // one serial chain per wave: issue ahead of the parse kernels' waves sharing the SIMD
    uint16_t* state = srstate + (size_t)c * kSrWords;
    const uint16_t* sin = state_in ? state_in + (size_t)c * kSrWords : state;
    const uint32_t base = (uint32_t)(uintptr_t)lds;  // (LDS byte address of value[0])
    // value[] holds a symbol as the LDS address of its index[] slot: rebuilding index[] is a store through the entry
    auto slot_of = [&](uint32_t sym) -> uint32_t { return base + kSrIdxOff + 2 * sym; };
    auto sym_of = [&](uint32_t slot) -> uint32_t { return (slot - base - kSrIdxOff) >> 1; };
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    auto at = [&](uint32_t slot) -> lds_u16& { return *reinterpret_cast<lds_u16*>((uintptr_t)slot); };  // (no index arithmetic)
    for (uint32_t i = lane; i < kSrPad; i += 64) {
        val[i] = (uint16_t)slot_of(i < kSyms ? sin[i] : i);
        idx[i] = i < kSyms ? sin[kSyms + i] : (uint16_t)i;
    }
    uint32_t cnt = __builtin_amdgcn_readfirstlane((int)(sin[2 * kSyms] | ((uint32_t)sin[2 * kSyms + 1] << 16)));
    uint32_t sum = __builtin_amdgcn_readfirstlane((int)(sin[2 * kSyms + 2] | ((uint32_t)sin[2 * kSyms + 3] << 16)));
    __syncthreads();
    bool idx_ok = true;  // groups do not maintain index[]
    auto rebuild_idx = [&]() {  // seven reads in flight, then seven writes
        uint32_t s[7];
#pragma unroll
        for (uint32_t m = 0; m < 7; m++) s[m] = val[lane + 64 * m];
#pragma unroll
        for (uint32_t m = 0; m < 7; m++) at(s[m]) = (uint16_t)(lane + 64 * m);
    };
    auto load_pairs = [&](uint32_t j0) -> uint32_t {  // lanes 2k and 2k+1: item j0 + k
        return j0 + kSrGroup <= e ? gsym[j0 + (lane >> 1)] : 0u;
    };
    // src/symrank.rs:43-47: the excluded symbol's rank is skipped; the symbol itself being the excluded one codes as the last rank
    auto out_rank = [&](uint32_t ri, uint32_t ru) -> uint16_t { return (uint16_t)(ri == ru ? kSyms - 1 : ri - (ri > ru)); };
    const uint32_t l2 = base + kSrSnapOff + 2 * lane;
    const uint32_t* const snap_pair = reinterpret_cast<const uint32_t*>(lds + kSrSnapOff / 2 + (lane & 31) * 66);  // item k: lanes 2k, 2k+1 of snapshot k
    const uint64_t m01 = 1;  // lane 0
    uint32_t j = a;
    while (j < e) {
        if (cnt >= 192 && e - j >= kSrGroup) {
            // ---- groups of 32 items while there are 32
            uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)((sum >> 4) / cnt));
            uint32_t pairs = load_pairs(j);
            do {
                uint32_t s[7];  // value[] as the group finds it: index[] is rebuilt from it, and a failed speculation restores it
#pragma unroll
                for (uint32_t m = 0; m < 7; m++) s[m] = val[lane + 64 * m];
#pragma unroll
                for (uint32_t m = 0; m < 7; m++) at(s[m]) = (uint16_t)(lane + 64 * m);
                const uint32_t symslot = slot_of((lane & 1) ? pairs >> 16 : pairs & 0xffff);
                const int x0 = (int)at(symslot);
                pairs = load_pairs(j + kSrGroup);  // in flight while this group runs
                q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
                uint32_t si;
                uint64_t m1, m3;
                int x, vi, t, h, nx, y, rr, rw, ar, aw, n;
                // In the steady state (count 352..390, a small quotient) the group runs WITHOUT the scalar chain, on the
                // assumption that the quotient stays q through its 32 items -- across the one scaling by 9/10 that may fall
                // into it, whose place is known from the count (src/symrank.rs:63-66) -- and the assumption is checked
                // afterwards, in every lane for one item: count and sum after each item follow from the 32 ranks by a prefix
                // sum.  Where it fails (once in ~60 groups on text) value[] is put back and the group runs again, checked.
                bool done = false;
                if (cnt >= 327 && q < 32) {
                    x = x0;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                                 ORZ_SRL_GROUP_S
                                 "s_waitcnt lgkmcnt(0)\n\tds_write_b16 %[aw], %[n]\n\t"
                                 "v_lshl_add_u32 %[ar], %[x], 1, %[base]\n\tds_write_b16 %[ar], %[sym]\n\t"
                                 : [x] "+v"(x), [si] "=&s"(si), [m1] "=&s"(m1), [m3] "=&s"(m3), [vi] "=&v"(vi), [t] "=&v"(t), [h] "=&v"(h),
                                   [nx] "=&v"(nx), [y] "=&v"(y), [rr] "=&v"(rr), [rw] "=&v"(rw), [ar] "=&v"(ar), [aw] "=&v"(aw), [n] "=&v"(n)
                                 : [l2] "v"(l2), [base] "s"(base), [sym] "v"(symslot), [m01] "s"(m01), [c15] "s"(15 - (int)(q << 4))
                                 : "vcc", "memory");
                    const uint32_t two = *snap_pair;
                    // inclusive prefix sum of the items' ranks over lanes 0..31 (two rows of sixteen)
                    uint32_t ps = lane < kSrGroup ? two & 0xffff : 0;
                    ps += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ps, 0x111, 0xf, 0xf, true);  // row_shr:1
                    ps += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ps, 0x112, 0xf, 0xf, true);  // row_shr:2
                    ps += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ps, 0x114, 0xf, 0xf, true);  // row_shr:4
                    ps += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ps, 0x118, 0xf, 0xf, true);  // row_shr:8
                    ps += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ps, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1 and 3
                    const uint32_t r = kSyms + 1 - cnt;  // the item that starts with count 390 scales first (none: r >= 32)
                    const uint32_t pr = r > 0 && r < kSrGroup ? (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)((r - 1) & 31)) : 0u;
                    const uint32_t scaled = (sum + pr) * 9 / 10;
                    const bool after = lane >= r;
                    const uint32_t cnt_k = after ? (kSyms + 1) * 9 / 10 + (lane - r) + 1 : cnt + lane + 1;
                    const uint32_t sum_k = after ? scaled + (ps - pr) : sum + ps;
                    const uint32_t lo = (q << 4) * cnt_k;
                    const bool ok = sum_k >= lo && sum_k - lo < (cnt_k << 4);
                    if (__ballot(lane < kSrGroup && !ok) == 0) {
                        if (lane < kSrGroup) grank[j + lane] = out_rank(two & 0xffff, two >> 16);
                        cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_k, 31);
                        sum = (uint32_t)__builtin_amdgcn_readlane((int)sum_k, 31);
                        done = true;
                    } else {
#pragma unroll
                        for (uint32_t m = 0; m < 7; m++) val[lane + 64 * m] = (uint16_t)s[m];
                    }
                }
                if (!done) {
                    uint32_t s0, s1;
                    uint32_t qw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cnt << 4));
                    uint32_t qc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q << 4));
                    uint32_t qa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sum - q * qw));
                    x = x0;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                                 ORZ_SRL_GROUP
                                 // the last item's displaced values; then every tracked symbol to where it ended up
                                 "s_waitcnt lgkmcnt(0)\n\tds_write_b16 %[aw], %[n]\n\t"
                                 "v_lshl_add_u32 %[ar], %[x], 1, %[base]\n\tds_write_b16 %[ar], %[sym]\n\t"
                                 : [x] "+v"(x), [qw] "+s"(qw), [qa] "+s"(qa), [qc] "+s"(qc), [q] "+s"(q), [si] "=&s"(si), [s0] "=&s"(s0),
                                   [s1] "=&s"(s1), [m1] "=&s"(m1), [m3] "=&s"(m3), [vi] "=&v"(vi), [t] "=&v"(t), [h] "=&v"(h), [nx] "=&v"(nx),
                                   [y] "=&v"(y), [rr] "=&v"(rr), [rw] "=&v"(rw), [ar] "=&v"(ar), [aw] "=&v"(aw), [n] "=&v"(n)
                                 : [l2] "v"(l2), [base] "s"(base), [sym] "v"(symslot), [m01] "s"(m01)
                                 : "scc", "vcc", "memory");
                    cnt = qw >> 4;
                    sum = qa + q * qw;
                    const uint32_t two = *snap_pair;
                    if (lane < kSrGroup) grank[j + lane] = out_rank(two & 0xffff, two >> 16);
                }
                j += kSrGroup;
            } while (e - j >= kSrGroup);
            idx_ok = false;
            continue;
        }
        // ---- one item the plain way (both tables in LDS)
        if (!idx_ok) { rebuild_idx(); idx_ok = true; }
        const uint32_t g = __builtin_amdgcn_readfirstlane((int)gsym[j]);
        const uint32_t v = g & 0xffff, vun = g >> 16;
        const uint32_t i = idx[v], iu = idx[vun];
        if (cnt > kSyms) {  // src/symrank.rs:63-66
            cnt = cnt * 9 / 10;
            sum = sum * 9 / 10;
        }
        cnt += 1;
        sum += i;
        const uint32_t dec = (i >> 4) + (sum >> 4) / cnt, half = i >> 1;
        uint32_t nx = i > dec ? i - dec : 0;
        nx = nx > half ? nx : half;
        const uint32_t y = (i + nx) >> 1;
        // value[ni1] <- value[next_i], value[i] <- value[ni1], value[next_i] <- v in this order: a swap (ni1 == next_i) and no
        // move (all three equal) come out right without a branch (src/symrank.rs:75-96)
        const uint32_t nv1 = val[y], nv2 = val[nx];
        val[y] = (uint16_t)nv2; at(nv2) = (uint16_t)y;
        val[i] = (uint16_t)nv1; at(nv1) = (uint16_t)i;
        val[nx] = (uint16_t)slot_of(v); idx[v] = (uint16_t)nx;
        if (lane == 0) grank[j] = out_rank(i, iu);
        j += 1;
    }
    // tables back to HBM
    if (!idx_ok) rebuild_idx();
    __syncthreads();
    for (uint32_t i = lane; i < kSyms; i += 64) { state[i] = (uint16_t)sym_of(val[i]); state[kSyms + i] = idx[i]; }
    if (lane == 0) {
        state[2 * kSyms] = (uint16_t)cnt;
        state[2 * kSyms + 1] = (uint16_t)(cnt >> 16);
        state[2 * kSyms + 2] = (uint16_t)sum;
        state[2 * kSyms + 3] = (uint16_t)(sum >> 16);
    }
}

#endif  // __HIPCC__

}  // namespace orz
