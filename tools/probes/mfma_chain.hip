// MFMA issue-pattern probe: 8 accumulators (a 128x64 wave tile), 48 v_mfma_f32_32x32x16_bf16 per iteration, either as
// 6-long dependent chains per accumulator (the bf16x6 order) or round-robin over the accumulators; 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE> __global__ void __launch_bounds__(512) k(float *out, int iters, const bf16x8 *in)
{
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
    bf16x8 a[4][3], b[2][3];
    for (int i = 0; i < 4; ++i) for (int p = 0; p < 3; ++p) a[i][p] = in[threadIdx.x + 64 * (i * 3 + p)];
    for (int j = 0; j < 2; ++j) for (int p = 0; p < 3; ++p) b[j][p] = in[threadIdx.x + 64 * (12 + j * 3 + p)];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i * 2 + j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i * 2 + j] = c;
                }
        } else {
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i * 2 + j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out; bf16x8 *in;
    hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&in, 64 * 32 * 16); hipMemset(in, 0x3c, 64 * 32 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode)
            for (int threads = 256; threads <= 512; threads += 256) {
                for (int w = 0; w < 2; ++w) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, in);
                    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, in);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double mf = 256.0 * (threads / 64) * iters * 48, tf = mf * 32 * 32 * 16 * 2 / ms / 1e9;
                printf("mode %s waves/SIMD %d: %.3f ms  %.0f TF  (%.1f ns per MFMA per SIMD)\n", mode ? "round-robin" : "chain-6   ", threads / 256, ms, tf,
                       ms * 1e6 / (iters * 48.0 * (threads / 256)));
            }
    return 0;
}
