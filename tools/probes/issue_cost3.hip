// issue_cost3.hip -- third pass: every 8-instruction group is ONE asm statement (hipcc's hazard recognizer pads s_nop between
// separate asm statements, which polluted issue_cost2), plus the exact mask patterns the composite kernels use.
//   build:  hipcc --offload-arch=gfx950 -O3 tools/probes/issue_cost3.hip -o build/issue_cost3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define G8(pre, post) pre "%0" post "\n\t" pre "%1" post "\n\t" pre "%2" post "\n\t" pre "%3" post "\n\t" pre "%4" post "\n\t" pre "%5" post "\n\t" pre "%6" post "\n\t" pre "%7" post
// two-register forms: dst %n, src %n
#define G8D(pre, mid, post) pre "%0" mid "%0" post "\n\t" pre "%1" mid "%1" post "\n\t" pre "%2" mid "%2" post "\n\t" pre "%3" mid "%3" post "\n\t" \
                            pre "%4" mid "%4" post "\n\t" pre "%5" mid "%5" post "\n\t" pre "%6" mid "%6" post "\n\t" pre "%7" mid "%7" post

#define OPS(X)                                                                                         \
    X(fma_vop3, 8, G8D("v_fma_f32 ", ", ", ", %8, %9"))                                                \
    X(fmac, 8, G8("v_fmac_f32 ", ", %8, %9"))                                                          \
    X(mul, 8, G8D("v_mul_f32 ", ", ", ", %8"))                                                         \
    X(add_sgpr_src, 8, G8D("v_add_f32 ", ", %12, ", ""))                                               \
    X(min, 8, G8D("v_min_f32 ", ", ", ", %8"))                                                         \
    X(max, 8, G8D("v_max_f32 ", ", ", ", %8"))                                                         \
    X(med3, 8, G8D("v_med3_f32 ", ", ", ", %8, %9"))                                                   \
    X(cmp_vcc, 8, G8("v_cmp_gt_f32 vcc, ", ", %8"))                                                    \
    X(cmp_sgpr, 8, G8("v_cmp_gt_f32_e64 %10, ", ", %8"))                                               \
    X(cmp_class, 8, G8("v_cmp_lt_u32 vcc, ", ", %8"))                                                  \
    X(cndmask_vcc_stale, 8, G8D("v_cndmask_b32_e32 ", ", ", ", %8, vcc"))                              \
    X(cndmask_sgpr, 8, G8D("v_cndmask_b32_e64 ", ", ", ", %8, %10"))                                   \
    X(cndmask_sgpr_zero, 8, G8D("v_cndmask_b32_e64 ", ", 0, ", ", %10"))                               \
    X(salu_and_vcc_then_cndmask, 24, G8D("s_and_b64 vcc, %10, %11\n\ts_nop 1\n\tv_cndmask_b32_e32 ", ", 0, ", ", vcc"))  \
    X(salu_and_sgpr_then_cndmask, 24, G8D("s_and_b64 %10, %10, %11\n\ts_nop 1\n\tv_cndmask_b32_e64 ", ", 0, ", ", %10")) \
    X(vcmp_then_cndmask_vcc, 16, G8D("v_cmp_gt_f32 vcc, %8, %9\n\tv_cndmask_b32_e32 ", ", 0, ", ", vcc"))                \
    X(vcmp_sgpr_then_cndmask, 16, G8D("v_cmp_gt_f32_e64 %10, %8, %9\n\tv_cndmask_b32_e64 ", ", 0, ", ", %10"))           \
    X(vcmp_and_then_cndmask, 32, G8D("v_cmp_gt_f32 vcc, %8, %9\n\ts_and_b64 vcc, vcc, %11\n\ts_nop 1\n\tv_cndmask_b32_e32 ", ", 0, ", ", vcc")) \
    X(mul_by_mask01, 8, G8D("v_mul_f32 ", ", ", ", %9"))                                               \
    X(lshlrev, 8, G8D("v_lshlrev_b32 ", ", 1, ", ""))                                                  \
    X(cvt_f32_i32, 8, G8D("v_cvt_f32_i32 ", ", ", ""))                                                 \
    X(mad_u32_u24, 8, G8D("v_mad_u32_u24 ", ", ", ", %8, %9"))                                         \
    X(add3_u32, 8, G8D("v_add3_u32 ", ", ", ", %8, %9"))                                               \
    X(dpp_quad_perm, 8, G8D("v_add_f32_dpp ", ", %8, ", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")) \
    X(dpp_row_ror, 8, G8D("v_add_f32_dpp ", ", %8, ", " row_ror:4 row_mask:0xf bank_mask:0xf"))        \
    X(dpp_row_bcast15, 8, G8D("v_add_f32_dpp ", ", %8, ", " row_bcast:15 row_mask:0xa bank_mask:0xf")) \
    X(dpp_mov, 8, G8("v_mov_b32_dpp ", ", %8 row_ror:8 row_mask:0xf bank_mask:0xf"))                   \
    X(swap32, 8, "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7") \
    X(readfirstlane, 8, G8("v_readfirstlane_b32 %12, ", ""))                                           \
    X(readlane, 8, G8("v_readlane_b32 %12, ", ", 5"))                                                  \
    X(s_nop0, 8, "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0")     \
    X(v_nop, 8, "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop")               \
    X(salu_and64, 8, "s_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11\n\ts_and_b64 %10, %10, %11") \
    X(fma4_salu4, 8, "v_fmac_f32 %0, %8, %9\n\ts_and_b64 %10, %10, %11\n\tv_fmac_f32 %1, %8, %9\n\ts_and_b64 %10, %10, %11\n\tv_fmac_f32 %2, %8, %9\n\ts_and_b64 %10, %10, %11\n\tv_fmac_f32 %3, %8, %9\n\ts_and_b64 %10, %10, %11") \
    X(exp, 8, G8D("v_exp_f32 ", ", ", ""))                                                             \
    X(fmac6_exp1_rcp1, 8, "v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_exp_f32 %6, %6\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\tv_rcp_f32 %7, %7\n\tv_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9")

enum OpId {
#define X(name, n, str) OP_##name,
    OPS(X)
#undef X
    OP_COUNT
};
static const char *NAMES[] = {
#define X(name, n, str) #name,
    OPS(X)
#undef X
};
static const int NINST[] = {
#define X(name, n, str) n,
    OPS(X)
#undef X
};

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 0.999f, c = 1e-3f;
    unsigned long long m = 0x5555555555555555ull, m2 = 0xffffffff0000ffffull;
    float sc = 0.25f;
    asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            switch (OP) {
#define X(name, n, str) case OP_##name: asm volatile(str : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b), "+v"(c), "+s"(m), "+s"(m2), "+s"(sc) : : "vcc", "scc"); break;
                OPS(X)
#undef X
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + c + sc + (float)(m + m2);
}

template <int OP>
static void run(float *out, int ncu)
{
    const int iters = (OP == OP_cndmask_vcc_stale) ? 256 : 2048;
    printf("%-30s", NAMES[OP]);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = ncu * w;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  w=%d: %7.3f", w, ms * 1e6 / ((double)iters * 4 * NINST[OP] * w));
    }
    printf("   ns per wave-instruction per SIMD (%d instr per group)\n", NINST[OP]);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 1000;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, (size_t)ncu * 8 * 256 * 4);
#define X(name, n, str) if (OP_##name >= first && OP_##name <= last) run<OP_##name>(out, ncu);
    OPS(X)
#undef X
    return 0;
}
