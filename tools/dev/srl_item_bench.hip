// dev: the item of the lanes formulation of the symbol-ranking chain (orz_amd/csrc/orz_symrank.h) in isolation -- variants of
// its schedule and ablations, one wavefront, a loop of 16 unrolled items over a fake table; ns and cycles per item.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/srl_item_bench.hip -o build/srl_item_bench && build/srl_item_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
constexpr int kIters = 1 << 16;

#define SIDE(P)                                                                                                          \
    P "30:\n\t"                                                                                                          \
    "s_lshr_b32 %[s0], %[qw], 4\n\ts_mul_i32 %[s1], %[s0], %[qc]\n\ts_add_u32 %[s1], %[s1], %[qa]\n\ts_mul_i32 %[s0], %[s0], 9\n\t" \
    "s_mul_hi_u32 %[s0], %[s0], 0xcccccccd\n\ts_lshr_b32 %[s0], %[s0], 3\n\ts_mul_i32 %[s1], %[s1], 9\n\t"              \
    "s_mul_hi_u32 %[s1], %[s1], 0xcccccccd\n\ts_lshr_b32 %[s1], %[s1], 3\n\ts_lshl_b32 %[qw], %[s0], 4\n\t"            \
    "s_mul_i32 %[s0], %[s0], %[qc]\n\ts_sub_u32 %[qa], %[s1], %[s0]\n\ts_branch " P "31b\n\t"                            \
    P "40:\n\t"                                                                                                          \
    "s_cmp_lt_i32 %[qa], 0\n\ts_cbranch_scc1 " P "42f\n\ts_sub_u32 %[qa], %[qa], %[qw]\n\ts_add_u32 %[q], %[q], 1\n\t"   \
    "s_add_u32 %[qc], %[qc], 16\n\ts_branch " P "43f\n\t"                                                                \
    P "42:\n\t"                                                                                                          \
    "s_add_u32 %[qa], %[qa], %[qw]\n\ts_sub_u32 %[q], %[q], 1\n\ts_sub_u32 %[qc], %[qc], 16\n\t"                        \
    P "43:\n\t"                                                                                                          \
    "s_cmp_ge_u32 %[qa], %[qw]\n\ts_cbranch_scc1 " P "40b\n\ts_branch " P "41b\n\t"

// ---- variant A: the first schedule (write-back at the item's start, compares into SGPR pairs, one x)
#define ITEM_A(P, J2, SNAP, AC, AP)                                                                                      \
    "ds_write_b16 %[l2], %[xA] offset:" SNAP "\n\t"                                                                      \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\t"                                                                            \
    "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\t"                               \
    "v_mov_b32 %[vi], %[si]\n\tv_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "s_cmp_ge_u32 %[qw], 0x1860\n\ts_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                              \
    "s_add_u32 %[qw], %[qw], 16\n\ts_add_u32 %[qa], %[qa], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"                                                                            \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\t"      \
    "v_lshl_add_u32 %[aiA], %[vi], 1, %[base]\n\tv_add_u32 %[y], %[vi], %[nx]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\t" \
    "v_lshrrev_b32 %[y], 1, %[y]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t"  \
    "v_lshl_add_u32 %[ayA], %[y], 1, %[base]\n\tds_read_u16 %[n2], %[anx]\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t" \
    "ds_read_u16 %[n1], %[ayA]\n\tv_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"

// ---- variant C: the third schedule (compares through VCC, x double-buffered, write-back mid-item)
#define ITEM_C(P, J2, SNAP, AC, AP)                                                                                      \
    "ds_write_b16 %[l2], %[x" AP "] offset:" SNAP "\n\t"                                                                 \
    "v_readlane_b32 %[si], %[x" AP "], " J2 "\n\t"                                                                       \
    "s_cmp_ge_u32 %[qw], 0x1860\n\ts_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                              \
    "s_add_u32 %[qw], %[qw], 16\n\tv_mov_b32 %[vi], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\tv_lshrrev_b32 %[t], 4, %[vi]\n\t" \
    "s_add_u32 %[qa], %[qa], %[si]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"                                                                            \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\tv_add_u32 %[y], %[vi], %[nx]\n\t"              \
    "v_cmp_eq_u32_e32 vcc, %[x" AP "], %[nx]\n\tv_lshrrev_b32 %[y], 1, %[y]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\t" \
    "v_cndmask_b32_e32 %[x" AC "], %[x" AP "], %[y], vcc\n\tv_cmp_eq_u32_e32 vcc, %[x" AP "], %[y]\n\t"                  \
    "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[ai" AP "], %[n1]\n\t"                                                        \
    "v_cndmask_b32_e32 %[x" AC "], %[x" AC "], %[vi], vcc\n\tv_cmp_eq_u32_e32 vcc, %[x" AP "], %[vi]\n\t"                \
    "ds_write_b16 %[ay" AP "], %[n2]\n\tv_lshl_add_u32 %[ay" AC "], %[y], 1, %[base]\n\t"                                \
    "v_cndmask_b32_e32 %[x" AC "], %[x" AC "], %[nx], vcc\n\tds_read_u16 %[n2], %[anx]\n\tds_read_u16 %[n1], %[ay" AC "]\n\t" \
    "v_lshl_add_u32 %[ai" AC "], %[vi], 1, %[base]\n\t"

// ---- ablations of A
#define ITEM_A_NOLDS(P, J2, SNAP, AC, AP)                                                                                \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\t"                                                                            \
    "v_mov_b32 %[vi], %[si]\n\tv_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "s_cmp_ge_u32 %[qw], 0x1860\n\ts_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                              \
    "s_add_u32 %[qw], %[qw], 16\n\ts_add_u32 %[qa], %[qa], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"                                                                            \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\t"      \
    "v_add_u32 %[y], %[vi], %[nx]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\t"                                           \
    "v_lshrrev_b32 %[y], 1, %[y]\n\ts_nop 0\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t"                                    \
    "s_nop 1\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t"                                                         \
    "v_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"
#define ITEM_A_NOQ(P, J2, SNAP, AC, AP)                                                                                  \
    "ds_write_b16 %[l2], %[xA] offset:" SNAP "\n\t"                                                                      \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\t"                                                                            \
    "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\t"                               \
    "v_mov_b32 %[vi], %[si]\n\tv_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\t"      \
    "v_lshl_add_u32 %[aiA], %[vi], 1, %[base]\n\tv_add_u32 %[y], %[vi], %[nx]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\t" \
    "v_lshrrev_b32 %[y], 1, %[y]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t"  \
    "v_lshl_add_u32 %[ayA], %[y], 1, %[base]\n\tds_read_u16 %[n2], %[anx]\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t" \
    "ds_read_u16 %[n1], %[ayA]\n\tv_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"
// only the map: readlane + 3 compares + 3 selects against fixed nx, y
#define ITEM_MAPONLY(P, J2, SNAP, AC, AP)                                                                                \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\ts_nop 1\n\tv_mov_b32 %[vi], %[si]\n\t"                                       \
    "v_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t" \
    "s_nop 1\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t"                                                         \
    "v_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"
// only the LDS traffic of an item
#define ITEM_LDSONLY(P, J2, SNAP, AC, AP)                                                                                \
    "ds_write_b16 %[l2], %[xA] offset:" SNAP "\n\ts_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\t" \
    "ds_read_u16 %[n2], %[anx]\n\tds_read_u16 %[n1], %[ayA]\n\t"
// only the scalar chain
#define ITEM_QONLY(P, J2, SNAP, AC, AP)                                                                                  \
    "s_cmp_ge_u32 %[qw], 0x1860\n\ts_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                              \
    "s_add_u32 %[qw], %[qw], 16\n\ts_add_u32 %[qa], %[qa], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"
// only the targets
#define ITEM_TGTONLY(P, J2, SNAP, AC, AP)                                                                                \
    "v_mov_b32 %[vi], %[si]\n\tv_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\tv_add_u32 %[y], %[vi], %[nx]\n\tv_lshrrev_b32 %[y], 1, %[y]\n\t" \
    "v_lshl_add_u32 %[aiA], %[vi], 1, %[base]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\tv_lshl_add_u32 %[ayA], %[y], 1, %[base]\n\t"


// ---- variant D: A's registers, but every s_cbranch sits four or more instructions behind its s_cmp (the scalar chain alone
// costs 57 cycles with the pairs adjacent: an untaken branch on a fresh SCC waits)
#define ITEM_D(P, J2, SNAP, AC, AP)                                                                                      \
    "ds_write_b16 %[l2], %[xA] offset:" SNAP "\n\t"                                                                      \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\t"                                                                            \
    "s_cmpk_ge_u32 %[qw], 0x1860\n\t"                                                                                    \
    "s_waitcnt lgkmcnt(1)\n\tds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\t"                               \
    "v_mov_b32 %[vi], %[si]\n\t"                                                                                         \
    "s_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                                                            \
    "s_add_u32 %[qw], %[qw], 16\n\ts_add_u32 %[qa], %[qa], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" \
    "v_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_lshl_add_u32 %[aiA], %[vi], 1, %[base]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"                                                                            \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\t"      \
    "v_add_u32 %[y], %[vi], %[nx]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\t"                                           \
    "v_lshrrev_b32 %[y], 1, %[y]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t"  \
    "v_lshl_add_u32 %[ayA], %[y], 1, %[base]\n\tds_read_u16 %[n2], %[anx]\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t" \
    "ds_read_u16 %[n1], %[ayA]\n\tv_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"
// D with 32-bit LDS entries (value[] and the snapshots as dwords)
#define ITEM_D32(P, J2, SNAP, AC, AP)                                                                                    \
    "ds_write_b32 %[l2], %[xA] offset:" SNAP "\n\t"                                                                      \
    "v_readlane_b32 %[si], %[xA], " J2 "\n\t"                                                                            \
    "s_cmpk_ge_u32 %[qw], 0x1860\n\t"                                                                                    \
    "s_waitcnt lgkmcnt(1)\n\tds_write_b32 %[aiA], %[n1]\n\tds_write_b32 %[ayA], %[n2]\n\t"                               \
    "v_mov_b32 %[vi], %[si]\n\t"                                                                                         \
    "s_cbranch_scc1 " P "30f\n\t" P "31:\n\t"                                                                            \
    "s_add_u32 %[qw], %[qw], 16\n\ts_add_u32 %[qa], %[qa], %[si]\n\ts_sub_u32 %[qa], %[qa], %[qc]\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" \
    "v_lshrrev_b32 %[t], 4, %[vi]\n\tv_lshrrev_b32 %[h], 1, %[vi]\n\tv_lshl_add_u32 %[aiA], %[vi], 1, %[base]\n\tv_sub_u32 %[t], %[vi], %[t]\n\t" \
    "s_cbranch_scc1 " P "40f\n\t" P "41:\n\t"                                                                            \
    "v_sub_u32_e64 %[t], %[t], %[q]\n\tv_cmp_eq_u32_e64 %[m1], %[xA], %[vi]\n\tv_max3_i32 %[nx], %[t], %[h], 0\n\t"      \
    "v_add_u32 %[y], %[vi], %[nx]\n\tv_cmp_eq_u32_e64 %[m3], %[xA], %[nx]\n\t"                                           \
    "v_lshrrev_b32 %[y], 1, %[y]\n\tv_lshl_add_u32 %[anx], %[nx], 1, %[base]\n\tv_cmp_eq_u32_e32 vcc, %[xA], %[y]\n\t"  \
    "v_lshl_add_u32 %[ayA], %[y], 1, %[base]\n\tds_read_b32 %[n2], %[anx]\n\tv_cndmask_b32_e64 %[xA], %[xA], %[y], %[m3]\n\t" \
    "ds_read_b32 %[n1], %[ayA]\n\tv_cndmask_b32_e32 %[xA], %[xA], %[vi], vcc\n\tv_cndmask_b32_e64 %[xA], %[xA], %[nx], %[m1]\n\t"
// the scalar chain with fillers between each compare and its branch: K = 1, 2, 4 v_mov
#define ITEM_QF(K)                                                                                                       \
    "s_cmpk_ge_u32 %[qw], 0x1860\n\t" K "s_cbranch_scc1 9f\n\t"                                                          \
    "s_add_u32 %[qw], %[qw], 0\n\ts_add_u32 %[qa], %[qa], 0\n\ts_sub_u32 %[qa], %[qa], 0\n\ts_cmp_ge_u32 %[qa], %[qw]\n\t" K \
    "s_cbranch_scc1 9f\n\t"
#define F1 "v_mov_b32 %[t], %[vi]\n\t"
#define ITEM_QF0(P, J2, SNAP, AC, AP) ITEM_QF("")
#define ITEM_QF1(P, J2, SNAP, AC, AP) ITEM_QF(F1)
#define ITEM_QF2(P, J2, SNAP, AC, AP) ITEM_QF(F1 F1)
#define ITEM_QF4(P, J2, SNAP, AC, AP) ITEM_QF(F1 F1 F1 F1)
#define ITEM_F8(P, J2, SNAP, AC, AP) F1 F1 F1 F1 F1 F1 F1 F1
#define ITEM_W16(P, J2, SNAP, AC, AP) "ds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\tds_write_b16 %[aiA], %[n1]\n\tds_write_b16 %[ayA], %[n2]\n\t"
#define ITEM_W32(P, J2, SNAP, AC, AP) "ds_write_b32 %[aiA], %[n1]\n\tds_write_b32 %[ayA], %[n2]\n\tds_write_b32 %[aiA], %[n1]\n\tds_write_b32 %[ayA], %[n2]\n\t"
#define ITEM_W8(P, J2, SNAP, AC, AP) "ds_write_b8 %[aiA], %[n1]\n\tds_write_b8 %[ayA], %[n2]\n\tds_write_b8 %[aiA], %[n1]\n\tds_write_b8 %[ayA], %[n2]\n\t"
#define ITEM_WL32(P, J2, SNAP, AC, AP) "ds_write_b32 %[l2], %[xA]\n\tds_write_b32 %[l2], %[xA] offset:256\n\tds_write_b32 %[l2], %[xA]\n\tds_write_b32 %[l2], %[xA] offset:256\n\t"
#define ITEM_WL16(P, J2, SNAP, AC, AP) "ds_write_b16 %[l2], %[xA]\n\tds_write_b16 %[l2], %[xA] offset:256\n\tds_write_b16 %[l2], %[xA]\n\tds_write_b16 %[l2], %[xA] offset:256\n\t"

#define ITEMS16(I)                                                                                                       \
    I("100", "0", "0", "B", "A") I("101", "2", "64", "A", "B") I("102", "4", "128", "B", "A") I("103", "6", "192", "A", "B")      \
    I("104", "8", "256", "B", "A") I("105", "10", "320", "A", "B") I("106", "12", "384", "B", "A") I("107", "14", "448", "A", "B") \
    I("108", "16", "512", "B", "A") I("109", "18", "576", "A", "B") I("110", "20", "640", "B", "A") I("111", "22", "704", "A", "B") \
    I("112", "24", "768", "B", "A") I("113", "26", "832", "A", "B") I("114", "28", "896", "B", "A") I("115", "30", "960", "A", "B")
#define SIDES16 SIDE("100") SIDE("101") SIDE("102") SIDE("103") SIDE("104") SIDE("105") SIDE("106") SIDE("107") SIDE("108")  \
    SIDE("109") SIDE("110") SIDE("111") SIDE("112") SIDE("113") SIDE("114") SIDE("115")

#define KERNEL(name, ITEM, HALF)                                                                                         \
    __global__ __launch_bounds__(64) void name(int* out) {                                                               \
        __shared__ __attribute__((aligned(16))) uint16_t lds[2048];                                                      \
        const uint32_t lane = threadIdx.x, base = (uint32_t)(uintptr_t)lds;                                              \
        for (uint32_t i = lane; i < 448; i += 64) { lds[i] = (uint16_t)(base + 896 + 2 * i); lds[448 + i] = (uint16_t)i; } \
        __syncthreads();                                                                                                 \
        int xA = (int)((lane * 7) % 97), xB = 0;                                                                         \
        uint32_t q = 2, qw = 360 << 4, qc = q << 4, qa = 100, si = 5, s0, s1, it = kIters;                               \
        uint64_t m1, m3, ex;                                                                                             \
        int vi = 3, t, h, nx = 2, y = 2, aiA = base, ayA = base, aiB = base, ayB = base, anx = base, n1 = 0, n2 = 0;     \
        const uint32_t l2 = base + 1792 + 2 * lane;                                                                      \
        asm volatile("s_mov_b64 %[ex], exec\n\t" HALF "2:\n\t" ITEMS16(ITEM)                                             \
                     "s_sub_u32 %[it], %[it], 1\n\ts_cmp_lg_u32 %[it], 0\n\ts_cbranch_scc1 2b\n\ts_branch 9f\n\t" SIDES16  \
                     "9:\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %[ex]\n\t"                                           \
                     : [xA] "+v"(xA), [xB] "+v"(xB), [qw] "+s"(qw), [qa] "+s"(qa), [qc] "+s"(qc), [q] "+s"(q), [si] "+s"(si), \
                       [s0] "=&s"(s0), [s1] "=&s"(s1), [m1] "=&s"(m1), [m3] "=&s"(m3), [ex] "=&s"(ex), [vi] "+v"(vi), [t] "=&v"(t), \
                       [h] "=&v"(h), [nx] "+v"(nx), [y] "+v"(y), [aiA] "+v"(aiA), [ayA] "+v"(ayA), [aiB] "+v"(aiB),       \
                       [ayB] "+v"(ayB), [anx] "+v"(anx), [n1] "+v"(n1), [n2] "+v"(n2), [it] "+s"(it)                     \
                     : [l2] "v"(l2), [base] "s"(base)                                                                    \
                     : "scc", "vcc", "memory");                                                                          \
        out[lane] = xA + xB + q + qw + qa + qc + si + vi + nx + y + n1 + n2 + lds[lane];                                 \
    }
#define FULL ""
#define HALF32 "s_mov_b32 exec_hi, 0\n\t"
KERNEL(k_a_full, ITEM_A, FULL)
KERNEL(k_a_half, ITEM_A, HALF32)
KERNEL(k_c_full, ITEM_C, FULL)
KERNEL(k_c_half, ITEM_C, HALF32)
KERNEL(k_d_full, ITEM_D, FULL)
KERNEL(k_d32_full, ITEM_D32, FULL)
KERNEL(k_qf0, ITEM_QF0, FULL)
KERNEL(k_qf1, ITEM_QF1, FULL)
KERNEL(k_qf2, ITEM_QF2, FULL)
KERNEL(k_qf4, ITEM_QF4, FULL)
KERNEL(k_f8, ITEM_F8, FULL)
KERNEL(k_w16, ITEM_W16, FULL)
KERNEL(k_w32, ITEM_W32, FULL)
KERNEL(k_w8, ITEM_W8, FULL)
KERNEL(k_wl32, ITEM_WL32, FULL)
KERNEL(k_wl16, ITEM_WL16, FULL)
KERNEL(k_a_nolds, ITEM_A_NOLDS, FULL)
KERNEL(k_a_noq, ITEM_A_NOQ, FULL)
KERNEL(k_map, ITEM_MAPONLY, FULL)
KERNEL(k_map_half, ITEM_MAPONLY, HALF32)
KERNEL(k_lds, ITEM_LDSONLY, FULL)
KERNEL(k_lds_half, ITEM_LDSONLY, HALF32)
KERNEL(k_q, ITEM_QONLY, FULL)
KERNEL(k_tgt, ITEM_TGTONLY, FULL)
KERNEL(k_tgt_half, ITEM_TGTONLY, HALF32)

struct Test { const char* name; void (*fn)(int*); };
int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int* out; CHECK(hipMalloc(&out, 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Test> tests = {{"D full", k_d_full}, {"D full, 32-bit LDS entries", k_d32_full}, {"2 x (cmp, branch) + 3 scalar, no filler", k_qf0}, {"... 1 v_mov between", k_qf1}, {"... 2 v_mov between", k_qf2}, {"... 4 v_mov between", k_qf4}, {"8 v_mov", k_f8}, {"4 ds_write_b16 one address", k_w16}, {"4 ds_write_b32 one address", k_w32}, {"4 ds_write_b8 one address", k_w8}, {"4 ds_write_b32 lane addresses", k_wl32}, {"4 ds_write_b16 lane addresses", k_wl16},
        {"A full", k_a_full}, {"A half", k_a_half}, {"C full", k_c_full}, {"C half", k_c_half}, {"A without LDS ops", k_a_nolds},
        {"A without the scalar chain", k_a_noq}, {"map only", k_map}, {"map only, half", k_map_half}, {"LDS ops only", k_lds}, {"LDS ops only, half", k_lds_half},
        {"scalar chain only", k_q}, {"targets + addresses only", k_tgt}, {"targets + addresses only, half", k_tgt_half}};
    for (const Test& t : tests) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, out);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double ns = best * 1e6 / kIters / 16;
        printf("{\"variant\": \"%s\", \"ns_per_item\": %.2f, \"cycles_per_item_at_%.1fGHz\": %.1f}\n", t.name, ns, ghz, ns * ghz);
    }
    return 0;
}
