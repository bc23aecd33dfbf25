// tools/issue_microbench.hip -- development probe: what ONE wavefront alone on a SIMD pays per instruction on gfx950.
// The range coder's chains (csrc/ac_kernels.hip) are a single wavefront per stream; their cost model is this table.
//   hipcc --offload-arch=gfx950 -O2 tools/issue_microbench.hip -o /tmp/issue_microbench && /tmp/issue_microbench
// Every test runs REPS x 256 copies of a pattern inside one 64-thread block; ns per pattern from HIP events.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                              \
        }                                                                          \
    } while (0)

#define KERNEL(NAME, BODY)                                                         \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, int reps, uint32_t seed) { \
        uint32_t a = seed + threadIdx.x, b = seed * 3u + 1u, c = seed ^ 0x55u, d = seed + 7u; \
        uint32_t s0 = seed, s1 = seed + 1u, s2 = seed + 2u, s3 = seed + 3u;        \
        s0 = __builtin_amdgcn_readfirstlane(s0);                                   \
        s1 = __builtin_amdgcn_readfirstlane(s1);                                   \
        s2 = __builtin_amdgcn_readfirstlane(s2);                                   \
        s3 = __builtin_amdgcn_readfirstlane(s3);                                   \
        for (int r = 0; r < reps; ++r) {                                           \
            asm volatile(".rept 256\n" BODY "\n.endr"                               \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"vcc", "memory", "v100", "v101", "v102", "v103", "v104", "m0", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s90", "s91"); \
        }                                                                          \
        out[threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3;                      \
    }

// %0..%3 = VGPRs a..d, %4..%7 = SGPRs s0..s3
KERNEL(k_empty, "")
KERNEL(k_vadd_dep, "v_add_u32 %0, %0, %1")
KERNEL(k_vadd_ind, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
KERNEL(k_vmulhi_dep, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_vmulhi_ind, "v_mul_hi_u32 %0, %1, %4\n v_mul_hi_u32 %2, %3, %4")
KERNEL(k_vmul24_dep, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_vmad24_dep, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_vmad64_dep, "v_mad_u64_u32 v[100:101], vcc, %0, %1, v[100:101]\n v_mov_b32 %0, v100")
KERNEL(k_vmullo_dep, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_vffbh_dep, "v_ffbh_u32 %0, %0")
KERNEL(k_vdpp_dep, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1")
KERNEL(k_vsdwa_dep, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL(k_sadd_dep, "s_add_u32 %4, %4, %5")
KERNEL(k_sadd_ind, "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1")
KERNEL(k_slshl64_dep, "s_lshl_b64 s[90:91], s[90:91], 1")
KERNEL(k_snop, "s_nop 0")
KERNEL(k_cmp_bcnt, "v_cmp_ge_u32 vcc, %4, %0\n s_bcnt1_i32_b64 %4, vcc")                                   // S -> V -> S
KERNEL(k_cmp_bcnt4_vcc, "v_cmp_ge_u32 vcc, %4, %0\n s_bcnt1_i32_b64 %5, vcc\n v_cmp_ge_u32 vcc, %4, %1\n s_bcnt1_i32_b64 %6, vcc\n"
                        "v_cmp_ge_u32 vcc, %4, %2\n s_bcnt1_i32_b64 %7, vcc\n v_cmp_ge_u32 vcc, %4, %3\n s_bcnt1_i32_b64 %4, vcc")
KERNEL(k_cmp_bcnt4_sep, "v_cmp_ge_u32 s[80:81], %4, %0\n v_cmp_ge_u32 s[82:83], %4, %1\n v_cmp_ge_u32 s[84:85], %4, %2\n v_cmp_ge_u32 s[86:87], %4, %3\n"
                        "s_bcnt1_i32_b64 %5, s[80:81]\n s_bcnt1_i32_b64 %6, s[82:83]\n s_bcnt1_i32_b64 %7, s[84:85]\n s_bcnt1_i32_b64 %4, s[86:87]")
KERNEL(k_readlane_chain, "v_readlane_b32 %4, %0, %4")                                                        // S -> V -> S
KERNEL(k_readlane_sadd, "v_readlane_b32 %5, %0, %4\n s_add_u32 %4, %4, %5")
KERNEL(k_s_to_v, "s_add_u32 %4, %4, 1\n v_add_u32 %0, %0, %4")                                               // independent-ish
KERNEL(k_cndmask_vcc, "s_cmp_lg_u32 %4, 0\n s_cselect_b64 vcc, -1, 0\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_branch_taken, "s_branch 0")   // s_branch to the next instruction (simm16 = 0): a taken branch
KERNEL(k_branch_not_taken, "s_cmp_eq_u32 %4, %4\n s_cbranch_scc0 0")
KERNEL(k_gpr_idx, "s_set_gpr_idx_on %4, 1\n v_mov_b32 %0, v100\n s_set_gpr_idx_off")
KERNEL(k_gpr_idx2, "s_set_gpr_idx_on %4, 1\n v_mov_b32 %0, v100\n s_set_gpr_idx_idx %5\n v_mov_b32 %1, v100\n s_set_gpr_idx_off")
KERNEL(k_sel_cndmask, "s_bitcmp0_b32 %4, 6\n s_cselect_b64 vcc, -1, 0\n s_bitcmp0_b32 %4, 7\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n"
                      "s_cselect_b64 vcc, -1, 0\n v_cndmask_b32 %0, %2, %0, vcc")
KERNEL(k_readlane_x2_sadd, "v_readlane_b32 %5, %0, %4\n v_readlane_b32 %6, %1, %4\n s_add_u32 %4, %4, %5\n s_add_u32 %4, %4, %6")
KERNEL(k_writelane_m0, "s_mov_b32 m0, %4\n v_readlane_b32 %5, %1, %4\n v_writelane_b32 %0, %4, m0")
KERNEL(k_lds_chase, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)")

struct Test {
    const char *name;
    void (*fn)(uint32_t *, int, uint32_t);
    int per_pattern;   // instructions in one pattern
};

int main() {
    uint32_t *out;
    CHECK(hipMalloc(&out, 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Test tests[] = {
        {"(loop overhead only)", k_empty, 0},
        {"v_add_u32 dependent", k_vadd_dep, 1},
        {"v_add_u32 x4 independent", k_vadd_ind, 4},
        {"v_mul_hi_u32 dependent", k_vmulhi_dep, 1},
        {"v_mul_hi_u32 x2 independent", k_vmulhi_ind, 2},
        {"v_mul_u32_u24 dependent", k_vmul24_dep, 1},
        {"v_mad_u32_u24 dependent", k_vmad24_dep, 1},
        {"v_mad_u64_u32 + v_mov dependent", k_vmad64_dep, 2},
        {"v_mul_lo_u32 dependent", k_vmullo_dep, 1},
        {"v_ffbh_u32 dependent", k_vffbh_dep, 1},
        {"v_mov_dpp + s_nop 1 dependent", k_vdpp_dep, 2},
        {"v_add_u32_sdwa dependent", k_vsdwa_dep, 1},
        {"s_add_u32 dependent", k_sadd_dep, 1},
        {"s_add_u32 x4 independent", k_sadd_ind, 4},
        {"s_lshl_b64 dependent", k_slshl64_dep, 1},
        {"s_nop 0", k_snop, 1},
        {"v_cmp -> s_bcnt1 -> (v_cmp operand) chain", k_cmp_bcnt, 2},
        {"4 x (v_cmp vcc, s_bcnt1 vcc) as the compiler emits them", k_cmp_bcnt4_vcc, 8},
        {"4 x v_cmp into 4 SGPR pairs, then 4 x s_bcnt1", k_cmp_bcnt4_sep, 8},
        {"v_readlane -> lane select of the next (S->V->S chain)", k_readlane_chain, 1},
        {"v_readlane + s_add chain", k_readlane_sadd, 2},
        {"s_add + v_add reading it", k_s_to_v, 2},
        {"s_cmp, s_cselect vcc, v_cndmask chain", k_cndmask_vcc, 3},
        {"s_branch (taken, to the next instruction)", k_branch_taken, 1},
        {"s_cmp + s_cbranch not taken", k_branch_not_taken, 2},
        {"s_set_gpr_idx_on, v_mov, s_set_gpr_idx_off", k_gpr_idx, 3},
        {"s_set_gpr_idx_on, v_mov, s_set_gpr_idx_idx, v_mov, _off", k_gpr_idx2, 5},
        {"register select by 2 s_bitcmp, 2 s_cselect, 3 v_cndmask", k_sel_cndmask, 7},
        {"2 x v_readlane, then 2 x s_add reading them (chain)", k_readlane_x2_sadd, 4},
        {"s_mov m0, v_readlane, v_writelane ... m0", k_writelane_m0, 3},
        {"ds_read_b32 pointer chase + wait", k_lds_chase, 2},
    };
    const int reps = 400;
    std::printf("%-62s %10s %12s\n", "pattern (one wavefront alone, 256 x %d copies)", "ns/pattern", "ns/instr");
    double base = 0.0;
    for (const Test &t : tests) {
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(1), dim3(64), 0, 0, out, reps, 0u);   // seed 0: the LDS chase reads address 0 (holds what it holds)
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double ns = (double)best * 1e6 / (reps * 256.0);
        if (t.per_pattern == 0) base = ns;
        std::printf("%-62s %10.2f %12.2f\n", t.name, ns, t.per_pattern ? (ns - 0.0) / t.per_pattern : 0.0);
    }
    std::printf("(the empty pattern = launch + loop overhead per 256 copies: %.3f ns per copy, not subtracted)\n", base);
    return 0;
}
