// issue_cost.hip -- what one wave64 instruction costs a gfx950 SIMD, per opcode and per resident-wave count.
// The composite kernels (K5 / K6) are instruction-bound; this table is what their instruction budgets are priced with.
//   build:  hipcc --offload-arch=gfx950 -O3 tools/probes/issue_cost.hip -o build/issue_cost
//   run:    build/issue_cost            (prints cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA, FMAC, MUL, ADD, MIN, CNDMASK, CMP, AND, MOV, EXP, RCP, DPP_ADD, SWAP32, SWAP16, PKFMA, READLANE, SALU, LDSB128, LDSB32,
          FMA_DEP, EXP_DEP, MIX_EXP7, BPERMUTE, NOPS };
static const char *NAMES[] = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_cndmask_b32", "v_cmp_gt_f32",
                              "v_and_b32", "v_mov_b32", "v_exp_f32", "v_rcp_f32", "v_add_f32 dpp row_ror", "v_permlane32_swap",
                              "v_permlane16_swap", "v_pk_fma_f32", "v_readlane_b32", "s_add_u32", "ds_read_b128 (bcast)",
                              "ds_read_b32 (bcast)", "v_fma_f32 1 chain", "v_exp_f32 1 chain", "7 fma + 1 exp", "ds_bpermute_b32"};

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float4 s_buf[64];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 0.999f, c = 1e-3f;
    float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    if (threadIdx.x < 64) s_buf[threadIdx.x] = make_float4(a0, a1, a2, a3);
    __syncthreads();
    unsigned lds_addr = 0, bp = (threadIdx.x * 4 + 4) & 255;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (OP == FMA) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == FMAC) {
#define X(n) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == MUL) {
#define X(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == ADD) {
#define X(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##n) : "v"(c));
                REP8(X)
#undef X
            } else if (OP == MIN) {
#define X(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == CNDMASK) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(b) : );
                REP8(X)
#undef X
            } else if (OP == CMP) {
#define X(n) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a##n), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (OP == AND) {
#define X(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == MOV) {
#define X(n) asm volatile("v_mov_b32 %0, %1" : "+v"(a##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == EXP) {
#define X(n) asm volatile("v_exp_f32 %0, %0" : "+v"(a##n));
                REP8(X)
#undef X
            } else if (OP == RCP) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##n));
                REP8(X)
#undef X
            } else if (OP == DPP_ADD) {
#define X(n) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a##n));
                REP8(X)
#undef X
            } else if (OP == SWAP32) {
                asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                             "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == SWAP16) {
                asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                             "v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\tv_permlane16_swap_b32 %4, %6\n\tv_permlane16_swap_b32 %5, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == PKFMA) {
                // 8 packed = 16 fp32 fma lanes-worth; counted as 8 instructions
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                             "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (OP == READLANE) {
#define X(n) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(a##n));
                REP8(X)
#undef X
            } else if (OP == SALU) {
#define X(n) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s2) : "scc");
                REP8(X)
#undef X
            } else if (OP == LDSB128) {
                float4 q;
#define X(n) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lds_addr)); a##n += q.x;
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == LDSB32) {
                float q;
#define X(n) asm volatile("ds_read_b32 %0, %1" : "=v"(q) : "v"(lds_addr)); a##n += q;
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == FMA_DEP) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == EXP_DEP) {
#define X(n) asm volatile("v_exp_f32 %0, %0" : "+v"(a0));
                REP8(X)
#undef X
            } else if (OP == MIX_EXP7) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_exp_f32 %7, %7\n\t"
                             "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == BPERMUTE) {
#define X(n) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##n) : "v"(bp));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + (float)(s0 + s1 + s3);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(float *out, unsigned long long *cyc, int ncu)
{
    const int iters = 4096, per_iter = 32;
    printf("%-26s", NAMES[OP]);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = ncu * w;   // 256-thread blocks: one wave per SIMD each, w blocks per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, cyc, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= h.size();
        // s_memtime ticks per instruction issued on one SIMD (w waves share it)
        const double ticks = avg / ((double)iters * per_iter * w);
        const double ns = ms * 1e6 / ((double)iters * per_iter * w);
        printf("  w=%d: %6.2f tick %6.3f ns", w, ticks, ns);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz; columns: w resident waves per SIMD; per wave-instruction PER SIMD: s_memtime ticks and wall ns\n",
           p.gcnArchName, ncu, p.clockRate);
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)ncu * 8 * 256 * 4);
    hipMalloc(&cyc, (size_t)ncu * 8 * 4 * 8);
    run<FMA>(out, cyc, ncu); run<FMAC>(out, cyc, ncu); run<MUL>(out, cyc, ncu); run<ADD>(out, cyc, ncu); run<MIN>(out, cyc, ncu);
    run<CNDMASK>(out, cyc, ncu); run<CMP>(out, cyc, ncu); run<AND>(out, cyc, ncu); run<MOV>(out, cyc, ncu); run<EXP>(out, cyc, ncu);
    run<RCP>(out, cyc, ncu); run<DPP_ADD>(out, cyc, ncu); run<SWAP32>(out, cyc, ncu); run<SWAP16>(out, cyc, ncu); run<PKFMA>(out, cyc, ncu);
    run<READLANE>(out, cyc, ncu); run<SALU>(out, cyc, ncu); run<LDSB128>(out, cyc, ncu); run<LDSB32>(out, cyc, ncu);
    run<FMA_DEP>(out, cyc, ncu); run<EXP_DEP>(out, cyc, ncu); run<MIX_EXP7>(out, cyc, ncu); run<BPERMUTE>(out, cyc, ncu);
    return 0;
}
