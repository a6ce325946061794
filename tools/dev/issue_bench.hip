// dev: what does ONE wavefront pay per instruction on gfx950?  The symbol-ranking chain (orz_symrank_kernel) is a lone wave
// per context running ~30 scalar / lane-access instructions per item; this measures the building blocks of that loop in
// isolation: cycles per instruction for dependent / independent SALU and VALU streams, the VALU -> SGPR -> SALU and
// VALU -> SGPR -> lane-select hops, v_readlane / v_writelane pairs, a taken branch.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench [GHz=2.4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int kIters = 1 << 20;

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// every kernel: one wave, kIters trips of a body of `n` instructions of interest (+ s_sub, s_cmp, s_cbranch = 3 of loop)
#define KERNEL(name, body, clobbers...)                                                                   \
    __global__ __launch_bounds__(64) void name(int* out) {                                                  \
        int v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = 7, v3 = 9;                                         \
        int s0 = 1, s1 = 2, s2 = 3, s3 = 5;                                                                 \
        unsigned it = kIters;                                                                               \
        asm volatile("s_mov_b32 m0, 3\n\t"                                                                  \
                     "1:\n\t" body                                                                          \
                     "s_sub_u32 %[it], %[it], 1\n\ts_cmp_lg_u32 %[it], 0\n\ts_cbranch_scc1 1b\n\t"         \
                     : [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [it] "+s"(it) \
                     :                                                                                      \
                     : "scc", "vcc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", ##clobbers);           \
        out[threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;                                           \
    }

KERNEL(k_empty, "")
KERNEL(k_salu_dep8, REP8("s_add_u32 %[s0], %[s0], 1\n\t"))
KERNEL(k_salu_ind8, REP4("s_add_u32 %[s0], %[s0], 1\n\ts_add_u32 %[s1], %[s1], 1\n\t"))
KERNEL(k_valu_dep8, REP8("v_add_u32 %[v0], %[v0], 1\n\t"))
KERNEL(k_valu_ind8, REP4("v_add_u32 %[v0], %[v0], 1\n\tv_add_u32 %[v1], %[v1], 1\n\t"))
KERNEL(k_mix_ind8, REP4("v_add_u32 %[v0], %[v0], 1\n\ts_add_u32 %[s0], %[s0], 1\n\t"))
KERNEL(k_snop8, REP8("s_nop 0\n\t"))
// VALU compare -> SGPR pair -> s_ff1 -> (SALU) -> back into the VALU as an operand: 4 dependent hops of 3 instructions
KERNEL(k_cmp_ff1_x4, REP4("v_cmp_eq_u32 s[40:41], %[v0], %[v1]\n\ts_ff1_i32_b64 %[s0], s[40:41]\n\tv_add_u32 %[v0], %[s0], %[v0]\n\t"))
// the same with four independent instructions between the compare and its consumer
KERNEL(k_cmp_4_ff1_x4, REP4("v_cmp_eq_u32 s[40:41], %[v0], %[v1]\n\ts_add_u32 %[s1], %[s1], 1\n\ts_add_u32 %[s2], %[s2], 1\n\ts_add_u32 %[s3], %[s3], 1\n\ts_add_u32 %[s1], %[s1], 1\n\t"
                             "s_ff1_i32_b64 %[s0], s[40:41]\n\tv_add_u32 %[v0], %[s0], %[v0]\n\t"))
// v_readlane (SGPR result) -> v_writelane using it as the value: dependent pairs
KERNEL(k_rl_wl_x4, REP4("v_readlane_b32 s42, %[v0], 5\n\ts_nop 0\n\tv_writelane_b32 %[v0], s42, m0\n\t"))
// v_readlane whose result is the lane select of the next v_readlane (4 wait states required), result written back
KERNEL(k_rl_sel_rl_x2, REP4("v_readlane_b32 s42, %[v2], 5\n\ts_nop 3\n\tv_readlane_b32 s43, %[v0], s42\n\ts_nop 0\n\tv_writelane_b32 %[v2], s43, m0\n\t"))
// s_mov m0 + v_writelane pairs (independent values)
KERNEL(k_m0_wl_x4, REP4("s_mov_b32 m0, %[s1]\n\tv_writelane_b32 %[v0], %[s2], m0\n\t"))
// a taken branch in the body (besides the loop's own)
KERNEL(k_branch_x4, REP4("s_cmp_eq_u32 %[s0], %[s0]\n\ts_cbranch_scc1 2f\n\ts_nop 0\n\t2:\n\t") )
// independent readlanes
KERNEL(k_rl_ind8, REP4("v_readlane_b32 s42, %[v0], 5\n\tv_readlane_b32 s43, %[v1], 6\n\t"))
// sdwa compare as the kernel uses it
KERNEL(k_sdwa_cmp_ind4, REP4("v_cmp_eq_u32_sdwa s[40:41], %[v0], %[v1] src0_sel:WORD_0 src1_sel:WORD_0\n\t"))

// compare variants: where does the mask go, how wide is the compare
KERNEL(k_cmp_e64_ind4, REP4("v_cmp_eq_u32_e64 s[40:41], %[s0], %[v1]\n\t"))
KERNEL(k_cmp_u16_e64_ind4, REP4("v_cmp_eq_u16_e64 s[40:41], %[s0], %[v1]\n\t"))
KERNEL(k_cmp_vcc_ind4, REP4("v_cmp_eq_u32_e32 vcc, %[s0], %[v1]\n\t"))
KERNEL(k_cmp_u16_vcc_ind4, REP4("v_cmp_eq_u16_e32 vcc, %[s0], %[v1]\n\t"))
KERNEL(k_cmp_sdwa_vcc_ind4, REP4("v_cmp_eq_u32_sdwa vcc, %[s0], %[v1] src0_sel:WORD_1 src1_sel:WORD_0\n\t"))
// compare into VCC, branch on it (never taken: v0 != s0 somewhere... all lanes differ), find the bit
KERNEL(k_cmp_vccz_ff1_x4, REP4("v_cmp_ne_u32_e32 vcc, %[s3], %[v1]\n\ts_cbranch_vccz 3f\n\ts_ff1_i32_b64 %[s0], vcc\n\t") "3:\n\t")
KERNEL(k_cmp_ff1_cmp_br_x4, REP4("v_cmp_ne_u32_e64 s[40:41], %[s3], %[v1]\n\ts_ff1_i32_b64 %[s0], s[40:41]\n\ts_cmp_lt_i32 %[s0], 0\n\ts_cbranch_scc1 3f\n\t") "3:\n\t")
// untaken branches
KERNEL(k_untaken_x8, REP8("s_cbranch_scc1 3f\n\t") "3:\n\t")
KERNEL(k_readlane_sgprsel_ind8, REP4("v_readlane_b32 s42, %[v0], %[s1]\n\tv_readlane_b32 s43, %[v1], %[s2]\n\t"))
KERNEL(k_readfirstlane_ind8, REP4("v_readfirstlane_b32 s42, %[v0]\n\tv_readfirstlane_b32 s43, %[v1]\n\t"))
KERNEL(k_writelane_const_ind8, REP8("v_writelane_b32 %[v0], %[s2], 5\n\t"))
KERNEL(k_writelane_m0_ind8, REP8("v_writelane_b32 %[v0], %[s2], m0\n\t"))
KERNEL(k_smov_m0_8, REP8("s_mov_b32 m0, %[s1]\n\t"))


// ---- round 6: what the lanes formulation of the chain is made of (orz_symrank.h)
#define KERNEL_X(name, pre, body, post)                                                                     \
    __global__ __launch_bounds__(64) void name(int* out) {                                                  \
        __shared__ int lds[1024];                                                                           \
        lds[threadIdx.x] = threadIdx.x;                                                                     \
        __syncthreads();                                                                                    \
        int v0 = threadIdx.x, v1 = threadIdx.x * 3, v2 = 7, v3 = 9;                                         \
        int va = (int)(uintptr_t)lds + 2 * threadIdx.x;                                                     \
        int s0 = 1, s1 = 2, s2 = 3, s3 = 5;                                                                 \
        unsigned it = kIters;                                                                               \
        asm volatile(pre "1:\n\t" body                                                                      \
                     "s_sub_u32 %[it], %[it], 1\n\ts_cmp_lg_u32 %[it], 0\n\ts_cbranch_scc1 1b\n\t" post    \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                             \
                     : [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [it] "+s"(it) \
                     : [va] "v"(va)                                                                         \
                     : "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "memory");                   \
        out[threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + lds[threadIdx.x];                        \
    }
#define EXEC32 "s_mov_b64 s[44:45], exec\n\ts_mov_b32 exec_hi, 0\n\t"
#define EXEC16 "s_mov_b64 s[44:45], exec\n\ts_mov_b32 exec_hi, 0\n\ts_mov_b32 exec_lo, 0xffff\n\t"
#define EXECBACK "s_mov_b64 exec, s[44:45]\n\t"
KERNEL_X(k_valu8_full, "", REP8("v_add_u32 %[v0], %[v0], 1\n\t"), "")
KERNEL_X(k_valu8_exec32, EXEC32, REP8("v_add_u32 %[v0], %[v0], 1\n\t"), EXECBACK)
KERNEL_X(k_valu8_exec16, EXEC16, REP8("v_add_u32 %[v0], %[v0], 1\n\t"), EXECBACK)
KERNEL_X(k_cmpvcc4_exec32, EXEC32, REP4("v_cmp_eq_u32_e32 vcc, %[s0], %[v1]\n\t"), EXECBACK)
KERNEL_X(k_cndmask8, "", REP8("v_cndmask_b32_e32 %[v0], %[v0], %[v1], vcc\n\t"), "")
KERNEL_X(k_cndmask8_e64, "", REP8("v_cndmask_b32_e64 %[v0], %[v0], %[v1], s[40:41]\n\t"), "")
KERNEL_X(k_max3_8, "", REP8("v_max3_i32 %[v0], %[v0], %[v1], 0\n\t"), "")
KERNEL_X(k_lshladd8, "", REP8("v_lshl_add_u32 %[v0], %[v1], 1, %[s1]\n\t"), "")
KERNEL_X(k_vmov_s8, "", REP8("v_mov_b32 %[v0], %[s1]\n\t"), "")
KERNEL_X(k_dswrite8, "", REP8("ds_write_b16 %[va], %[v1]\n\t"), "")
KERNEL_X(k_dswrite8_off, "", REP4("ds_write_b16 %[va], %[v1] offset:128\n\tds_write_b16 %[va], %[v1] offset:256\n\t"), "")
KERNEL_X(k_dsread8, "", REP8("ds_read_u16 %[v2], %[va]\n\t"), "")
KERNEL_X(k_dsread_wait4, "", REP4("ds_read_u16 %[v2], %[va]\n\ts_waitcnt lgkmcnt(0)\n\t"), "")
KERNEL_X(k_dswrite8_uniform, "v_mov_b32 %[v3], 64\n\t", REP8("ds_write_b16 %[v3], %[v1]\n\t"), "")
KERNEL_X(k_dsread8_uniform, "v_mov_b32 %[v3], 64\n\t", REP8("ds_read_u16 %[v2], %[v3]\n\t"), "")
KERNEL_X(k_waitcnt8, "", REP8("s_waitcnt lgkmcnt(1)\n\t"), "")
KERNEL_X(k_mix_vs8, "", REP4("v_add_u32 %[v0], %[v0], 1\n\ts_add_u32 %[s0], %[s0], 1\n\tv_add_u32 %[v1], %[v1], 1\n\ts_add_u32 %[s1], %[s1], 1\n\t"), "")
KERNEL_X(k_smax8, "", REP8("s_max_i32 %[s0], %[s0], %[s1]\n\t"), "")
KERNEL_X(k_slshl1add8, "", REP8("s_lshl1_add_u32 %[s0], %[s0], %[s1]\n\t"), "")
KERNEL_X(k_readlane_vmov4, "", REP4("v_readlane_b32 s42, %[v0], 6\n\ts_nop 1\n\tv_mov_b32 %[v1], s42\n\t"), "")
KERNEL_X(k_bpermute4, "", REP4("ds_bpermute_b32 %[v2], %[v3], %[v0]\n\ts_waitcnt lgkmcnt(0)\n\t"), "")
KERNEL_X(k_dpp_bcast4, "", REP4("v_mov_b32_dpp %[v2], %[v0] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"), "")

struct Test { const char* name; void (*fn)(int*); int n; };

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int* out;
    CHECK(hipMalloc(&out, 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Test> tests = {
        {"loop alone (s_sub, s_cmp, s_cbranch taken)", k_empty, 0},
        {"8 dependent s_add", k_salu_dep8, 8}, {"8 s_add, two chains", k_salu_ind8, 8},
        {"8 dependent v_add", k_valu_dep8, 8}, {"8 v_add, two chains", k_valu_ind8, 8},
        {"4 x (v_add, s_add) independent", k_mix_ind8, 8}, {"8 s_nop 0", k_snop8, 8},
        {"4 x (v_cmp -> s_ff1 -> v_add) dependent", k_cmp_ff1_x4, 12}, {"4 x (v_cmp, 4 s_add, s_ff1, v_add)", k_cmp_4_ff1_x4, 28},
        {"4 x (v_readlane, s_nop 0, v_writelane) dependent", k_rl_wl_x4, 12},
        {"4 x (v_readlane, s_nop 3, v_readlane by it, s_nop 0, v_writelane)", k_rl_sel_rl_x2, 20},
        {"4 x (s_mov m0, v_writelane)", k_m0_wl_x4, 8}, {"4 x (s_cmp, taken s_cbranch over one s_nop)", k_branch_x4, 8},
        {"8 independent v_readlane", k_rl_ind8, 8}, {"4 independent v_cmp_sdwa to an SGPR pair", k_sdwa_cmp_ind4, 4},
        {"4 v_cmp_eq_u32_e64 to an SGPR pair", k_cmp_e64_ind4, 4}, {"4 v_cmp_eq_u16_e64 to an SGPR pair", k_cmp_u16_e64_ind4, 4},
        {"4 v_cmp_eq_u32_e32 to VCC", k_cmp_vcc_ind4, 4}, {"4 v_cmp_eq_u16_e32 to VCC", k_cmp_u16_vcc_ind4, 4}, {"4 v_cmp_eq_u32_sdwa to VCC", k_cmp_sdwa_vcc_ind4, 4},
        {"4 x (v_cmp to VCC, s_cbranch_vccz untaken, s_ff1 vcc)", k_cmp_vccz_ff1_x4, 12},
        {"4 x (v_cmp to SGPRs, s_ff1, s_cmp_lt, s_cbranch untaken)", k_cmp_ff1_cmp_br_x4, 16},
        {"8 untaken s_cbranch_scc1", k_untaken_x8, 8}, {"8 v_readlane, lane select in an SGPR", k_readlane_sgprsel_ind8, 8},
        {"8 v_readfirstlane", k_readfirstlane_ind8, 8}, {"8 v_writelane, constant lane", k_writelane_const_ind8, 8},
        {"8 v_writelane, lane in M0", k_writelane_m0_ind8, 8}, {"8 s_mov m0", k_smov_m0_8, 8},
        {"8 v_add (kernel with LDS)", k_valu8_full, 8}, {"8 v_add, 32 lanes in EXEC", k_valu8_exec32, 8}, {"8 v_add, 16 lanes in EXEC", k_valu8_exec16, 8},
        {"4 v_cmp to VCC, 32 lanes in EXEC", k_cmpvcc4_exec32, 4}, {"8 v_cndmask by VCC", k_cndmask8, 8}, {"8 v_cndmask by an SGPR pair", k_cndmask8_e64, 8},
        {"8 v_max3_i32", k_max3_8, 8}, {"8 v_lshl_add_u32 with an SGPR", k_lshladd8, 8}, {"8 v_mov from an SGPR", k_vmov_s8, 8},
        {"8 ds_write_b16 (lane addresses)", k_dswrite8, 8}, {"8 ds_write_b16 with offsets", k_dswrite8_off, 8}, {"8 ds_read_u16", k_dsread8, 8},
        {"4 x (ds_read_u16, wait)", k_dsread_wait4, 8}, {"8 ds_write_b16, one address", k_dswrite8_uniform, 8}, {"8 ds_read_u16, one address", k_dsread8_uniform, 8},
        {"8 s_waitcnt lgkmcnt(1), nothing pending", k_waitcnt8, 8}, {"4 x (v_add, s_add, v_add, s_add)", k_mix_vs8, 16},
        {"8 s_max_i32", k_smax8, 8}, {"8 s_lshl1_add_u32", k_slshl1add8, 8}, {"4 x (v_readlane, s_nop 1, v_mov from it)", k_readlane_vmov4, 12},
        {"4 x (ds_bpermute, wait)", k_bpermute4, 8}, {"4 v_mov_dpp row_bcast", k_dpp_bcast4, 4},
    };
    double base = 0;
    for (const Test& t : tests) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double cyc = best * 1e6 * ghz / kIters;  // cycles per trip at `ghz`
        if (t.n == 0) base = cyc;
        printf("{\"test\": \"%s\", \"instructions\": %d, \"ns_per_trip\": %.2f, \"cycles_per_trip_at_%.1fGHz\": %.1f, \"cycles_per_instruction_net_of_loop\": %.2f}\n", t.name,
               t.n, best * 1e6 / kIters, ghz, cyc, t.n ? (cyc - base) / t.n : 0.0);
    }
    return 0;
}
