// vit_gemm.hip -- fp32 Linear with fused epilogue for the ViT blocks on gfx950.
//
//     out = [residual +] act( x (M,K) . w^T (N,K) + bias (N) ),   act in {identity, exact GELU}
//
// The reference computes these layers in fp32 (TF32 on NVIDIA, blocks.py:61-82,97-134); gfx950 has no
// reduced-precision fp32 path, so the contraction runs on the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 (157 TF peak = 1/16 of bf16).  128x128 output tile per workgroup, 4 wavefronts
// in a 2x2 arrangement of 64x64 sub-tiles (2x2 MFMA tiles of 32x32 each), K consumed 16 at a time
// through double-buffered LDS tiles stored K-MAJOR ([k][m] / [k][n]) so that both MFMA operand reads
// (lane = row) are unit-stride and conflict-free; the next K-slab is prefetched into registers while the
// current one feeds the MFMAs.  Epilogue in registers: + bias (lane = output column), exact erf GELU,
// + residual, optional second store of the pre-activation for the backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BK = 16, LDT = 132;   // LDT: padded row length of the k-major tiles (BN = 64 * TN)

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// TN = 32-column MFMA tiles per wavefront along N: TN = 2 -> 128x128 workgroup tile, TN = 1 -> 128x64 (more, smaller
// tiles for N <= 1024 layers, whose 128x128 tiling leaves most CUs with one tile while a few carry two)
template <int ACT, int TN>
__global__ void __launch_bounds__(256) k_linear(const float *__restrict__ x, const float *__restrict__ w,
                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                float *__restrict__ out, float *__restrict__ pre, int M, int N, int K)
{
    constexpr int BN = 64 * TN;
    __shared__ float sA[2][BK * LDT], sB[2][BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs, so give each XCD a
    // contiguous strip of tiles that share the same rows of x (L2 reuse)
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int bid = blockIdx.x;
    if (ntiles % 8 == 0) bid = (bid % 8) * (ntiles / 8) + bid / 8;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // loader mapping: BK/4 lanes cover one row segment of BK floats; 1024/BK rows per pass, BM*BK/1024 passes
    constexpr int LPR = BK / 4, RPP = 256 / LPR, NP = BM / RPP, NPB = BN / RPP;
    const int lrow = tid / LPR, lk = (tid % LPR) * 4;
    float4 ra[NP], rb[NPB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int gm = m0 + lrow + RPP * p;
            ra[p] = gm < M ? *reinterpret_cast<const float4 *>(x + (int64_t)gm * K + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p) {
            const int gn = n0 + lrow + RPP * p;
            rb[p] = gn < N ? *reinterpret_cast<const float4 *>(w + (int64_t)gn * K + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float *a = sA[buf] + lk * LDT + lrow + RPP * p;
            a[0] = ra[p].x; a[LDT] = ra[p].y; a[2 * LDT] = ra[p].z; a[3 * LDT] = ra[p].w;
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p) {
            float *b = sB[buf] + lk * LDT + lrow + RPP * p;
            b[0] = rb[p].x; b[LDT] = rb[p].y; b[2 * LDT] = rb[p].z; b[3 * LDT] = rb[p].w;
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float *a = sA[buf] + half * LDT + wm * 64 + col;   // A[m = wm*64 + 32 i + col][k = 2 s + half]
        const float *b = sB[buf] + half * LDT + wn * (32 * TN) + col;   // W[n = wn*32*TN + 32 j + col][k]
        // fragment reads run one k-step ahead of the MFMAs that consume them
        float a0 = a[0], a1 = a[32], b0 = b[0], b1 = TN > 1 ? b[32] : 0.f;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (s + 1 < BK / 2) {
                na0 = a[2 * (s + 1) * LDT]; na1 = a[2 * (s + 1) * LDT + 32];
                nb0 = b[2 * (s + 1) * LDT];
                if (TN > 1) nb1 = b[2 * (s + 1) * LDT + 32];
            }
            // D[i = m][j = n]: A-operand = x rows, B-operand = W rows -> lane = output column n (coalesced row stores)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            if (TN > 1) acc[0][TN - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][TN - 1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            if (TN > 1) acc[1][TN - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][TN - 1], 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // acc[i][j]: lane column = n = n0 + wn*64 + 32 j + col ; register r = row m = m0 + wm*64 + 32 i + (r&3) + 8 (r>>2) + 4 half
    // -> every store instruction writes two 128-byte row segments; bias is lane-local
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (32 * TN) + 32 * j + col;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const int64_t o = (int64_t)m * N + n;
                float t = acc[i][j][r] + bv;
                if (pre) pre[o] = t;
                if (ACT == 1) t = gelu_exact(t);
                if (residual) t += residual[o];
                out[o] = t;
            }
        }
    }
}

int linear_fwd(const float *x, const float *w, const float *bias, const float *residual, float *out, float *pre, int M, int N,
               int K, int act, hipStream_t stream)
{
    if (!x || !w || !out) return VIT_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0 || act < 0 || act > 1) return VIT_EINVAL;
    // tile shape: 128x128 unless that leaves fewer than ~2.5 tiles per CU (256 CUs), then 128x64
    const int tm = (M + BM - 1) / BM;
    const bool narrow = tm * ((N + 127) / 128) < 640;
    const int tiles = tm * (narrow ? (N + 63) / 64 : (N + 127) / 128);
    (void)hipGetLastError();
#define VIT_LAUNCH_LINEAR(ACT, TN) hipLaunchKernelGGL((k_linear<ACT, TN>), dim3(tiles), dim3(256), 0, stream, x, w, bias, residual, out, pre, M, N, K)
    if (act == 1) { if (narrow) VIT_LAUNCH_LINEAR(1, 1); else VIT_LAUNCH_LINEAR(1, 2); }
    else { if (narrow) VIT_LAUNCH_LINEAR(0, 1); else VIT_LAUNCH_LINEAR(0, 2); }
#undef VIT_LAUNCH_LINEAR
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
