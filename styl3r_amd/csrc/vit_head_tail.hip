// vit_head_tail.hip -- the last two layers of every DPT head as ONE pass each way (gfx950, wave64).
//
// Replaces (reference heads/dpt_block.py:313-341):
//   'regression' head tail:  nn.ReLU(True) -> Conv2d(last_dim = 128, 3, kernel_size = 1)                      (dpt_block.py:319-320)
//   'gs_params'  head tail:  nn.ReLU(True) -> nn.Dropout(0.1) -> Conv2d(feature_dim = 256, 8 | 3 d_sh, 1)    (:337-339)
// at 256 x 256 pixels.  These 1x1 convolutions have 3 .. 8 output channels: no GEMM tile fits them, the work is one pass over the
// 128- / 256-channel activation (671 MB for ten 256^2 images) and nothing else -- HBM-bound by construction.  The library ran them as
// implicit GEMMs behind separate ReLU / Dropout passes (six passes over the activation per step and head); here
//
//   forward   y[b, co, p] = bias[co] + sum_c W[co, c] a(h[b, c, p]),   a(h) = keep(b, c, p) max(h, 0) / (1 - p_drop)
//             one read of h; keep() comes from the same counter-based generator as vit_relu_dropout_fwd (Philox-4x32-10 keyed by
//             (seed, flat element index / 4)), so no mask and no activated copy of h is ever written;
//   backward  dh[b, c, p] = a'(h) sum_co W[co, c] dy[b, co, p]        (one read of h, one write of dh)
//             dW[co, c]   = sum_{b,p} dy[b, co, p] a(h[b, c, p]),  db[co] = sum dy      (same pass: the CO partial products of a wave's
//             256 pixels meet in a reduce-scatter butterfly of 10 lane exchanges, then in LDS, then one global atomic per workgroup
//             and output element).
//
// Lane = 4 consecutive pixels (one 16-byte load per channel, 1 KiB contiguous per wavefront); the weights are wave-uniform (scalar
// loads).  Algorithmic bytes: forward 4 B C P + 4 CO P per image, backward 8 B C P + 4 CO P.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

namespace ht {
__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key)      // identical to vit_resample.hip's (the two must agree bit for bit)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}

// a(h) for four elements and the factor a'(h) (0 or scale) of each
template <bool DROP>
__device__ inline void activate(const float4 v, int64_t i4, uint32_t thresh, float scale, uint64_t seed, float4 &a, float4 &m)
{
    if (DROP) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i4, (uint32_t)(i4 >> 32), 0u, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        m.x = (r.x < thresh && v.x > 0.f) ? scale : 0.f;
        m.y = (r.y < thresh && v.y > 0.f) ? scale : 0.f;
        m.z = (r.z < thresh && v.z > 0.f) ? scale : 0.f;
        m.w = (r.w < thresh && v.w > 0.f) ? scale : 0.f;
    } else {
        m.x = v.x > 0.f ? 1.f : 0.f; m.y = v.y > 0.f ? 1.f : 0.f; m.z = v.z > 0.f ? 1.f : 0.f; m.w = v.w > 0.f ? 1.f : 0.f;
    }
    a = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
}

template <int CO, bool DROP>
__global__ void __launch_bounds__(256) k_head_tail_fwd(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ bias,
                                                       float *__restrict__ y, int B, int C, int64_t HW4, uint32_t thresh, float scale, uint64_t seed)
{
    const int64_t total = (int64_t)B * HW4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t b = idx / HW4, q = idx - b * HW4;
        const float4 *hp = reinterpret_cast<const float4 *>(h) + b * C * HW4 + q;
        float4 acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) { const float bb = bias ? bias[co] : 0.f; acc[co] = make_float4(bb, bb, bb, bb); }
        for (int c0 = 0; c0 < C; c0 += 8) {             // eight channel rows in flight per lane
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = hp[(int64_t)(c0 + u) * HW4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4 a, m;
                activate<DROP>(v[u], (b * C + c0 + u) * HW4 + q, thresh, scale, seed, a, m);
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float ww = w[co * C + c0 + u];
                    acc[co].x = fmaf(ww, a.x, acc[co].x); acc[co].y = fmaf(ww, a.y, acc[co].y);
                    acc[co].z = fmaf(ww, a.z, acc[co].z); acc[co].w = fmaf(ww, a.w, acc[co].w);
                }
            }
        }
        float4 *yp = reinterpret_cast<float4 *>(y) + b * CO * HW4 + q;
#pragma unroll
        for (int co = 0; co < CO; ++co) yp[(int64_t)co * HW4] = acc[co];
    }
}

// Reduce-scatter butterfly: every lane enters with NV partial sums; afterwards the lanes of an 8-lane (NV = 8) or 16-lane (NV = 4)
// group all hold the wavefront total of ONE of the values: value index = lane >> 3 (NV = 8) / lane >> 4 (NV = 4).
template <int NV>
__device__ inline float wave_reduce_scatter(float (&v)[NV], int lane)
{
    static_assert(NV == 8 || NV == 4, "NV");
    if (NV == 8) {
        float k4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = (lane & 32) ? v[i] : v[4 + i];
            const float keep = (lane & 32) ? v[4 + i] : v[i];
            k4[i] = keep + __shfl_xor(send, 32);
        }
        float k2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = (lane & 16) ? k4[i] : k4[2 + i];
            const float keep = (lane & 16) ? k4[2 + i] : k4[i];
            k2[i] = keep + __shfl_xor(send, 16);
        }
        const float send = (lane & 8) ? k2[0] : k2[1];
        float r = ((lane & 8) ? k2[1] : k2[0]) + __shfl_xor(send, 8);
        r += __shfl_xor(r, 4); r += __shfl_xor(r, 2); r += __shfl_xor(r, 1);
        return r;
    } else {
        float k2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = (lane & 32) ? v[i] : v[2 + i];
            const float keep = (lane & 32) ? v[2 + i] : v[i];
            k2[i] = keep + __shfl_xor(send, 32);
        }
        const float send = (lane & 16) ? k2[0] : k2[1];
        float r = ((lane & 16) ? k2[1] : k2[0]) + __shfl_xor(send, 16);
        r += __shfl_xor(r, 8); r += __shfl_xor(r, 4); r += __shfl_xor(r, 2); r += __shfl_xor(r, 1);
        return r;
    }
}

constexpr int C_MAX = 256;

template <int CO, bool DROP>
__global__ void __launch_bounds__(256) k_head_tail_bwd(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ dy,
                                                       float *__restrict__ dh, float *__restrict__ dw, float *__restrict__ db, int B, int C,
                                                       int64_t HW4, uint32_t thresh, float scale, uint64_t seed)
{
    constexpr int NV = CO > 4 ? 8 : 4;                  // values per reduction (CO padded to a power of two)
    constexpr int GRP = 64 / NV;                        // lanes that end up with the same value
    __shared__ float s_dw[NV * C_MAX];
    __shared__ float s_db[NV];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < NV * C_MAX; i += 256) s_dw[i] = 0.f;
    if (tid < NV) s_db[tid] = 0.f;
    __syncthreads();
    float dbp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) dbp[i] = 0.f;
    const int64_t total = (int64_t)B * HW4;
    const int64_t rounds = (total + (int64_t)gridDim.x * 256 - 1) / ((int64_t)gridDim.x * 256);
    for (int64_t it = 0; it < rounds; ++it) {           // every lane takes part in every round (the butterflies need the whole wave)
        const int64_t idx = (it * gridDim.x + blockIdx.x) * 256 + tid;
        const bool live = idx < total;
        const int64_t b = live ? idx / HW4 : 0, q = live ? idx - b * HW4 : 0;
        float4 g[CO];
        const float4 *gp = reinterpret_cast<const float4 *>(dy) + b * CO * HW4 + q;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            g[co] = live ? gp[(int64_t)co * HW4] : make_float4(0.f, 0.f, 0.f, 0.f);
            dbp[co] += (g[co].x + g[co].y) + (g[co].z + g[co].w);
        }
        const float4 *hp = reinterpret_cast<const float4 *>(h) + b * C * HW4 + q;
        float4 *dp = reinterpret_cast<float4 *>(dh) + b * C * HW4 + q;
        for (int c0 = 0; c0 < C; c0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = live ? hp[(int64_t)(c0 + u) * HW4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                float4 a, m;
                activate<DROP>(v[u], (b * C + c) * HW4 + q, thresh, scale, seed, a, m);
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                float part[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) part[i] = 0.f;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float ww = w[co * C + c];
                    s.x = fmaf(ww, g[co].x, s.x); s.y = fmaf(ww, g[co].y, s.y); s.z = fmaf(ww, g[co].z, s.z); s.w = fmaf(ww, g[co].w, s.w);
                    part[co] = fmaf(g[co].x, a.x, fmaf(g[co].y, a.y, fmaf(g[co].z, a.z, g[co].w * a.w)));
                }
                if (live) dp[(int64_t)c * HW4] = make_float4(s.x * m.x, s.y * m.y, s.z * m.z, s.w * m.w);
                const float r = wave_reduce_scatter<NV>(part, lane);
                if ((lane & (GRP - 1)) == 0) atomicAdd(&s_dw[(lane / GRP) * C_MAX + c], r);
            }
        }
    }
    const float rb = wave_reduce_scatter<NV>(dbp, lane);
    if ((lane & (GRP - 1)) == 0) atomicAdd(&s_db[lane / GRP], rb);
    __syncthreads();
    for (int i = tid; i < CO * C; i += 256) {
        const int co = i / C, c = i - co * C;
        const float val = s_dw[co * C_MAX + c];
        if (val != 0.f) atomicAdd(dw + i, val);
    }
    if (db && tid < CO) atomicAdd(db + tid, s_db[tid]);
}
}  // namespace ht

static int ht_grid(int64_t total)
{
    const int64_t blocks = (total + 255) / 256;
    return (int)(blocks < 2048 ? blocks : 2048);       // 8 workgroups per CU resident; the rest by grid stride
}

int head_tail_fwd(const float *h, const float *w, const float *bias, float *y, int B, int C, int CO, int64_t HW, float p, uint64_t seed,
                  hipStream_t stream)
{
    if (!h || !w || !y || B <= 0 || C <= 0 || (C & 7) || C > ht::C_MAX || HW <= 0 || (HW & 3) || !(p >= 0.f && p < 1.f)) return VIT_EINVAL;
    if (CO != 3 && CO != 8) return VIT_EINVAL;
    const double keep = 1.0 - (double)p;
    const uint32_t thresh = (uint32_t)(keep * 4294967296.0 > 4294967295.0 ? 4294967295.0 : keep * 4294967296.0);
    const float scale = (float)(1.0 / keep);
    const int64_t HW4 = HW >> 2;
    const int grid = ht_grid((int64_t)B * HW4);
    (void)hipGetLastError();
#define VIT_HT_F(CO_, DROP_) hipLaunchKernelGGL((ht::k_head_tail_fwd<CO_, DROP_>), dim3(grid), dim3(256), 0, stream, h, w, bias, y, B, C, HW4, thresh, scale, seed)
    if (CO == 3) { if (p > 0.f) VIT_HT_F(3, true); else VIT_HT_F(3, false); }
    else { if (p > 0.f) VIT_HT_F(8, true); else VIT_HT_F(8, false); }
#undef VIT_HT_F
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int head_tail_bwd(const float *h, const float *w, const float *dy, float *dh, float *dw, float *db, int B, int C, int CO, int64_t HW, float p,
                  uint64_t seed, hipStream_t stream)
{
    if (!h || !w || !dy || !dh || !dw || B <= 0 || C <= 0 || (C & 7) || C > ht::C_MAX || HW <= 0 || (HW & 3) || !(p >= 0.f && p < 1.f)) return VIT_EINVAL;
    if (CO != 3 && CO != 8) return VIT_EINVAL;
    const double keep = 1.0 - (double)p;
    const uint32_t thresh = (uint32_t)(keep * 4294967296.0 > 4294967295.0 ? 4294967295.0 : keep * 4294967296.0);
    const float scale = (float)(1.0 / keep);
    const int64_t HW4 = HW >> 2;
    const int64_t total = (int64_t)B * HW4;
    int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    (void)hipGetLastError();
    if (hipMemsetAsync(dw, 0, (size_t)CO * C * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
    if (db && hipMemsetAsync(db, 0, (size_t)CO * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
#define VIT_HT_B(CO_, DROP_) hipLaunchKernelGGL((ht::k_head_tail_bwd<CO_, DROP_>), dim3(grid), dim3(256), 0, stream, h, w, dy, dh, dw, db, B, C, HW4, thresh, scale, seed)
    if (CO == 3) { if (p > 0.f) VIT_HT_B(3, true); else VIT_HT_B(3, false); }
    else { if (p > 0.f) VIT_HT_B(8, true); else VIT_HT_B(8, false); }
#undef VIT_HT_B
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
