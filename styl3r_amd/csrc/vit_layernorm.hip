// vit_layernorm.hip -- LayerNorm over the channel dimension of token tensors (croco/blocks.py:144-152,205-222:
// norm1 / norm2 / norm3 / norm_y of Block / DecoderBlock, enc_norm / dec_norm of the trunks; eps 1e-6), forward and
// backward, HBM-bound.
//
// One wavefront owns a row: C / 256 float4 per lane stay in registers, mean and variance are two exact wave
// reductions (two-pass: sum, then sum of squared deviations), no LDS, no workgroup barrier in the forward.
//
// Backward, one pass over (dy, x) per call:
//     dx = rstd * (g - mean_C(g) - xhat * mean_C(g * xhat)) [+ dskip],      g = dy * gamma, xhat = (x - mean) * rstd
// `dskip` is the gradient that reached the block's residual branch: every pre-norm block computes x + f(LN(x)), so
// the gradient of x is the sum of the skip gradient and the LayerNorm input gradient -- the framework would run a
// separate add kernel over the (M, C) tensor for it.  dgamma / dbeta: each lane keeps the column sums of its own
// columns in registers across the rows its workgroup visits, the four waves of a workgroup combine through LDS, every
// workgroup writes ONE partial row (coalesced), and k_ln_param_reduce adds the <= 512 partial rows (deterministic order,
// no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;
uint32_t *x6_take_output_amax();                                 // vit_gemm_x6.hip

namespace {
constexpr int LN_MAX_N4 = 8;        // C <= 2048
constexpr int LN_BWD_BLOCKS = 512;  // partial rows of the parameter gradients

__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ inline uint32_t abs_bits4(const float4 &o)
{
    return max(max(__builtin_bit_cast(uint32_t, o.x) & 0x7fffffffu, __builtin_bit_cast(uint32_t, o.y) & 0x7fffffffu),
               max(__builtin_bit_cast(uint32_t, o.z) & 0x7fffffffu, __builtin_bit_cast(uint32_t, o.w) & 0x7fffffffu));
}
__device__ inline void publish_amax(uint32_t *amax_out, uint32_t m, int lane)     // one guarded atomic per wave, spread over the 64-word line
{
    if (!amax_out) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    uint32_t *w = amax_out + ((blockIdx.x + (threadIdx.x >> 6)) & 63u) * 32;      // (64 slots, one per 128-byte line: vit_gemm_x6.hip AMAX_STRIDE)
    if (lane == 0 && m > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, m);
}

template <int N4>
__global__ void __launch_bounds__(256) k_ln_fwd(const float *__restrict__ x, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, float *__restrict__ y,
                                                float *__restrict__ mean, float *__restrict__ rstd, int M, float eps,
                                                uint32_t *__restrict__ amax_out)
{
    constexpr int C = N4 * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t omax = 0;          // |max| of the stored values, published for the f16x3 consumer (vit_x6_set_output_amax)
    float4 gm[N4], bt[N4];
#pragma unroll
    for (int j = 0; j < N4; ++j) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[j * 64 + lane];
        bt[j] = beta ? reinterpret_cast<const float4 *>(beta)[j * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
        float4 v[N4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) { v[j] = xr[j * 64 + lane]; s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
        const float mu = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float4 *yr = reinterpret_cast<float4 *>(y + (int64_t)row * C);
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const float4 o = make_float4(v[j].x * rs * gm[j].x + bt[j].x, v[j].y * rs * gm[j].y + bt[j].y,
                                         v[j].z * rs * gm[j].z + bt[j].z, v[j].w * rs * gm[j].w + bt[j].w);
            yr[j * 64 + lane] = o;
            omax = max(omax, abs_bits4(o));
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
    publish_amax(amax_out, omax, lane);
}

// Two LayerNorms of one shape in one launch (blockIdx.y = group): the dual decoders of the serving path (vit_layernorm_fwd_grouped).  Rows of
// group g: x_g / y_g with (gamma_g, beta_g); the |max| word covers both outputs.  No mean / rstd (forward only).
struct LnGroups { const float *x[2]; const float *gamma[2]; const float *beta[2]; float *y[2]; };
template <int N4>
__global__ void __launch_bounds__(256) k_ln_fwd_grouped(const LnGroups a, int M, float eps, uint32_t *__restrict__ amax_out)
{
    constexpr int C = N4 * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.y;
    const float *__restrict__ x = a.x[g], *__restrict__ gamma = a.gamma[g], *__restrict__ beta = a.beta[g];
    float *__restrict__ y = a.y[g];
    uint32_t omax = 0;
    float4 gm[N4], bt[N4];
#pragma unroll
    for (int j = 0; j < N4; ++j) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[j * 64 + lane];
        bt[j] = beta ? reinterpret_cast<const float4 *>(beta)[j * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {       // (the operations of k_ln_fwd)
        const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
        float4 v[N4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) { v[j] = xr[j * 64 + lane]; s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
        const float mu = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float4 *yr = reinterpret_cast<float4 *>(y + (int64_t)row * C);
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const float4 o = make_float4(v[j].x * rs * gm[j].x + bt[j].x, v[j].y * rs * gm[j].y + bt[j].y,
                                         v[j].z * rs * gm[j].z + bt[j].z, v[j].w * rs * gm[j].w + bt[j].w);
            yr[j * 64 + lane] = o;
            omax = max(omax, abs_bits4(o));
        }
    }
    publish_amax(amax_out, omax, lane);
}

template <int N4>
__global__ void __launch_bounds__(256) k_ln_bwd(const float *__restrict__ dy, const float *__restrict__ x,
                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                const float *__restrict__ gamma, const float *__restrict__ dskip,
                                                float *__restrict__ dx, float *__restrict__ partial, int M, uint32_t *__restrict__ amax_out)
{
    constexpr int C = N4 * 256;
    __shared__ float4 s_red[3][2 * N4 * 64];     // waves 1..3 hand their column sums to wave 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t omax = 0;
    float4 gm[N4], dg[N4], db[N4];
#pragma unroll
    for (int j = 0; j < N4; ++j) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[j * 64 + lane];
        dg[j] = db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
        const float4 *gr = reinterpret_cast<const float4 *>(dy + (int64_t)row * C);
        const float mu = mean[row], rs = rstd[row];
        float4 xh[N4], g[N4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const float4 xv = xr[j * 64 + lane], gv = gr[j * 64 + lane];
            xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            dg[j].x += gv.x * xh[j].x; dg[j].y += gv.y * xh[j].y; dg[j].z += gv.z * xh[j].z; dg[j].w += gv.w * xh[j].w;
            db[j].x += gv.x; db[j].y += gv.y; db[j].z += gv.z; db[j].w += gv.w;
            g[j] = make_float4(gv.x * gm[j].x, gv.y * gm[j].y, gv.z * gm[j].z, gv.w * gm[j].w);
            c1 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
            c2 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        }
        c1 = wave_sum(c1) * (1.0f / C);
        c2 = wave_sum(c2) * (1.0f / C);
        float4 *dr = reinterpret_cast<float4 *>(dx + (int64_t)row * C);
        const float4 *sr = dskip ? reinterpret_cast<const float4 *>(dskip + (int64_t)row * C) : nullptr;
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            float4 o = make_float4(rs * (g[j].x - c2 - xh[j].x * c1), rs * (g[j].y - c2 - xh[j].y * c1),
                                   rs * (g[j].z - c2 - xh[j].z * c1), rs * (g[j].w - c2 - xh[j].w * c1));
            if (sr) { const float4 sv = sr[j * 64 + lane]; o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w; }
            dr[j * 64 + lane] = o;
            omax = max(omax, abs_bits4(o));
        }
    }
    publish_amax(amax_out, omax, lane);
    // column sums of the workgroup -> partial[blockIdx.x][0..C) = dgamma, [C..2C) = dbeta
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < N4; ++j) { s_red[wave - 1][j * 64 + lane] = dg[j]; s_red[wave - 1][(N4 + j) * 64 + lane] = db[j]; }
    }
    __syncthreads();
    if (wave == 0) {
        float4 *pr = reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * 2 * C);
#pragma unroll
        for (int j = 0; j < N4; ++j) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 a = s_red[w][j * 64 + lane], b = s_red[w][(N4 + j) * 64 + lane];
                dg[j].x += a.x; dg[j].y += a.y; dg[j].z += a.z; dg[j].w += a.w;
                db[j].x += b.x; db[j].y += b.y; db[j].z += b.z; db[j].w += b.w;
            }
            pr[j * 64 + lane] = dg[j];
            pr[(N4 + j) * 64 + lane] = db[j];
        }
    }
}

// dgamma[c] (+)= sum_b partial[b][c], dbeta[c] (+)= sum_b partial[b][C + c].  A workgroup owns 32 columns; its 8 row groups
// each add every 8th partial row (8 independent loads in flight per thread: a thread-per-column loop over 512 rows is a
// 40 us chain of dependent-latency loads), then combine through LDS in a fixed order (deterministic, no atomics).
__global__ void __launch_bounds__(256) k_ln_param_reduce(const float *__restrict__ partial, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta, int nblk, int C, int accumulate)
{
    __shared__ float s_sum[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;                       // 2C is a multiple of 32
    const float *p = partial + c;
    const int64_t stride = 2 * (int64_t)C;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = rg;
    for (; b + 56 < nblk; b += 64) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(b + 8 * u) * stride];
    }
    for (; b < nblk; b += 8) acc[0] += p[(int64_t)b * stride];
    s_sum[rg][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (rg == 0) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += s_sum[r][cl];
        float *o = c < C ? dgamma + c : (dbeta ? dbeta + (c - C) : nullptr);
        if (o) *o = accumulate ? *o + s : s;
    }
}

int ln_blocks(int M) { const int b = (M + 3) / 4; return b < LN_BWD_BLOCKS ? b : LN_BWD_BLOCKS; }
}  // namespace

size_t layernorm_scratch_bytes(int M, int C) { return (size_t)ln_blocks(M) * 2 * (size_t)C * sizeof(float); }

int layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int M, int C,
                  float eps, hipStream_t stream)
{
    if (!x || !gamma || !y || !mean || !rstd || M <= 0 || C <= 0 || (C % 256) != 0 || C / 256 > LN_MAX_N4) return VIT_EINVAL;
    const int blocks = (M + 3) / 4 < 2048 ? (M + 3) / 4 : 2048;
    uint32_t *am_out = x6_take_output_amax();      // (vit_x6_set_output_amax: the |max| of y for an f16x3 consumer)
    (void)hipGetLastError();
#define VIT_LN_F(N4) case N4: hipLaunchKernelGGL(k_ln_fwd<N4>, dim3(blocks), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, eps, am_out); break
    switch (C / 256) { VIT_LN_F(1); VIT_LN_F(2); VIT_LN_F(3); VIT_LN_F(4); VIT_LN_F(5); VIT_LN_F(6); VIT_LN_F(7); VIT_LN_F(8); }
#undef VIT_LN_F
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int layernorm_fwd_grouped(const float *const *x, const float *const *gamma, const float *const *beta, float *const *y, int groups, int M, int C,
                          float eps, hipStream_t stream)
{
    if (!x || !gamma || !y || groups < 1 || groups > 2 || M <= 0 || C <= 0 || (C % 256) != 0 || C / 256 > LN_MAX_N4) return VIT_EINVAL;
    LnGroups a{};
    for (int g = 0; g < 2; ++g) {
        const int s = g < groups ? g : 0;
        if (!x[s] || !gamma[s] || !y[s]) return VIT_EINVAL;
        a.x[g] = x[s]; a.gamma[g] = gamma[s]; a.beta[g] = beta ? beta[s] : nullptr; a.y[g] = y[s];
    }
    const int blocks = (M + 3) / 4 < 2048 ? (M + 3) / 4 : 2048;
    uint32_t *am_out = x6_take_output_amax();
    (void)hipGetLastError();
#define VIT_LN_G(N4) case N4: hipLaunchKernelGGL(k_ln_fwd_grouped<N4>, dim3(blocks, groups), dim3(256), 0, stream, a, M, eps, am_out); break
    switch (C / 256) { VIT_LN_G(1); VIT_LN_G(2); VIT_LN_G(3); VIT_LN_G(4); VIT_LN_G(5); VIT_LN_G(6); VIT_LN_G(7); VIT_LN_G(8); }
#undef VIT_LN_G
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, const float *dskip,
                  float *dx, float *dgamma, float *dbeta, float *scratch, int M, int C, int accumulate, hipStream_t stream)
{
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !scratch || M <= 0 || (C % 256) != 0 || C / 256 > LN_MAX_N4)
        return VIT_EINVAL;
    const int blocks = ln_blocks(M);
    uint32_t *am_out = x6_take_output_amax();      // (the |max| of dx)
    (void)hipGetLastError();
#define VIT_LN_B(N4) case N4: hipLaunchKernelGGL(k_ln_bwd<N4>, dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, dskip, dx, scratch, M, am_out); break
    switch (C / 256) { VIT_LN_B(1); VIT_LN_B(2); VIT_LN_B(3); VIT_LN_B(4); VIT_LN_B(5); VIT_LN_B(6); VIT_LN_B(7); VIT_LN_B(8); }
#undef VIT_LN_B
    hipLaunchKernelGGL(k_ln_param_reduce, dim3(2 * C / 32), dim3(256), 0, stream, scratch, dgamma, dbeta, blocks, C, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
