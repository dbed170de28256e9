// issue_cost2.hip -- second table of per-opcode issue costs on gfx950 (follow-up of issue_cost.hip: v_cndmask_b32 measured
// 8x a v_fma_f32 there; this one separates encodings / operand kinds and adds the other opcodes the composite kernels use).
//   build:  hipcc --offload-arch=gfx950 -O3 tools/probes/issue_cost2.hip -o build/issue_cost2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// every op template acts on one accumulator %0 (VGPR, read-write); %1, %2 = VGPR inputs; %3 = SGPR input; %4 = SGPR pair (mask)
#define OPS(X)                                                                                          \
    X(fma_vvv, "v_fma_f32 %0, %0, %1, %2")                                                              \
    X(fma_sgpr, "v_fma_f32 %0, %0, %3, %2")                                                             \
    X(fma_neg_abs, "v_fma_f32 %0, -%0, |%1|, %2")                                                       \
    X(fmaak_literal, "v_fmaak_f32 %0, %0, %1, 0x3f7fbe77")                                              \
    X(mul_literal, "v_mul_f32 %0, 0x3f7fbe77, %0")                                                      \
    X(mul_inline_const, "v_mul_f32 %0, 0.5, %0")                                                        \
    X(add_sgpr, "v_add_f32 %0, %3, %0")                                                                 \
    X(sub, "v_sub_f32 %0, %0, %1")                                                                      \
    X(max, "v_max_f32 %0, %0, %1")                                                                      \
    X(min, "v_min_f32 %0, %0, %1")                                                                      \
    X(med3, "v_med3_f32 %0, %0, %1, %2")                                                                \
    X(cndmask_e32_vcc, "v_cndmask_b32_e32 %0, %0, %1, vcc")                                             \
    X(cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, %4")                                             \
    X(cndmask_dst_ne_src, "v_cndmask_b32_e32 %0, %1, %2, vcc")                                          \
    X(cndmask_zero_src, "v_cndmask_b32_e64 %0, 0, %0, %4")                                              \
    X(cmp_vcc, "v_cmp_gt_f32 vcc, %0, %1")                                                              \
    X(cmp_e64_sgpr, "v_cmp_gt_f32_e64 %4, %0, %1")                                                      \
    X(cmp_then_cndmask, "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")                \
    X(cmp_fma_fma_cndmask, "v_cmp_gt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc") \
    X(and_b32, "v_and_b32 %0, %0, %1")                                                                  \
    X(add_u32, "v_add_u32 %0, %0, %1")                                                                  \
    X(lshlrev, "v_lshlrev_b32 %0, 1, %0")                                                               \
    X(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")                                                              \
    X(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")                                                      \
    X(exp, "v_exp_f32 %0, %0")                                                                          \
    X(rcp, "v_rcp_f32 %0, %0")                                                                          \
    X(sqrt, "v_sqrt_f32 %0, %0")                                                                        \
    X(log, "v_log_f32 %0, %0")                                                                          \
    X(dpp_quad_perm, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")         \
    X(dpp_row_ror, "v_add_f32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf")                     \
    X(dpp_row_shr, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")                     \
    X(dpp_row_bcast15, "v_add_f32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")              \
    X(dpp_mov, "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf")                             \
    X(swap_b32, "v_swap_b32 %0, %1")                                                                    \
    X(readfirstlane, "v_readfirstlane_b32 %3, %0")                                                      \
    X(readlane, "v_readlane_b32 %3, %0, 5")                                                             \
    X(salu_and64, "s_and_b64 %4, %4, exec")                                                             \
    X(s_bcnt, "s_bcnt1_i32_b64 %3, %4")                                                                 \
    X(s_nop0, "s_nop 0")                                                                                \
    X(v_nop, "v_nop")                                                                                   \
    X(mfma_4x4x1, "v_mfma_f32_4x4x1_16b_f32 %5, %1, %2, %5")                                            \
    X(mfma_32x32x2_plus4fma, "v_mfma_f32_32x32x2_f32 %6, %1, %2, %6\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2")

enum OpId {
#define X(name, str) OP_##name,
    OPS(X)
#undef X
    OP_COUNT
};
static const char *NAMES[] = {
#define X(name, str) #name,
    OPS(X)
#undef X
};
static const int NINST[] = {   // wave-instructions per template expansion
#define X(name, str) 1,
    OPS(X)
#undef X
};

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 0.999f + threadIdx.x * 1e-6f, c = 1e-3f;
    float sc = 0.999f;
    unsigned long long m = 0x5555555555555555ull;
    f4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f16 acc16;
    for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
    asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            switch (OP) {
#define Y(n, str) asm volatile(str : "+v"(a##n), "+v"(b), "+v"(c), "+s"(sc), "+s"(m), "+v"(acc4), "+v"(acc16) : : "vcc");
#define X(name, str) case OP_##name: Y(0, str) Y(1, str) Y(2, str) Y(3, str) Y(4, str) Y(5, str) Y(6, str) Y(7, str) break;
                OPS(X)
#undef X
#undef Y
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + c + sc + (float)m + acc4[0] + acc16[0];
}

template <int OP>
static void run(float *out, int ncu, int mult)
{
    const int iters = 2048, per_iter = 32;
    printf("%-24s", NAMES[OP]);
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
        printf("  w=%d: %7.3f ns", w, ms * 1e6 / ((double)iters * per_iter * w * mult));
    }
    printf("   (per wave-instruction per SIMD; template = %d instr)\n", mult);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, (size_t)ncu * 8 * 256 * 4);
#define X(name, str) run<OP_##name>(out, ncu, OP_##name == OP_cmp_then_cndmask ? 2 : OP_##name == OP_cmp_fma_fma_cndmask ? 4 : OP_##name == OP_mfma_32x32x2_plus4fma ? 5 : 1);
    OPS(X)
#undef X
    (void)NINST;
    return 0;
}
