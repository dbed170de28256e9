// vit_resample.hip -- x2 bilinear up-sampling (align_corners = True) of the DPT heads
// (dpt_block.py: FeatureFusionBlock / the heads' Interpolate; F.interpolate(scale_factor=2, mode="bilinear",
// align_corners=True)).  The framework kernel writes these 0.25..1 GB outputs at ~0.35 TB/s; this one is a plain
// HBM-bound pass: one thread produces four consecutive output pixels (one 16-byte store), the 2 x <=4 input values
// it needs come from two input rows that stay in L1/L2 across the neighbouring threads.  Same formula and operation
// order as the framework (source index = dst * (in-1)/(out-1), lambda = fractional part), contraction off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) k_upsample2x(const float *__restrict__ in, float *__restrict__ out, int64_t planes, int H,
                                                    int W, float rh, float rw)
{
    const int OH = 2 * H, OW = 2 * W, OW4 = OW >> 2;
    const int64_t total = planes * OH * OW4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ox4 = (int)(idx % OW4);
        const int64_t t = idx / OW4;
        const int oy = (int)(t % OH);
        const int64_t pl = t / OH;
        const float h1r = rh * (float)oy;
        const int h1 = (int)h1r, h1p = (h1 < H - 1) ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
        const float *r0 = in + (pl * H + h1) * W, *r1 = r0 + (int64_t)h1p * W;
        float4 o;
        float *po = reinterpret_cast<float *>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = ox4 * 4 + e;
            const float w1r = rw * (float)ox;
            const int w1 = (int)w1r, w1p = (w1 < W - 1) ? 1 : 0;
            const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
            po[e] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
        }
        *reinterpret_cast<float4 *>(out + (pl * OH + oy) * OW + ox4 * 4) = o;
    }
}
// Backward as a GATHER (the framework scatters with atomics): input pixel (iy, ix) collects w_y * w_x * dout from the
// output rows / columns whose source cell touches it.  With scale (H-1)/(2H-1) < 1/2 those are among the six candidates
// 2i-2 .. 2i+3; each candidate's cell index is recomputed with the forward's own float arithmetic, so forward and
// backward agree on every floor() decision.  One pass: dout read once (cached across neighbours), din written once.
__global__ void __launch_bounds__(256) k_upsample2x_bwd(const float *__restrict__ dout, float *__restrict__ din, int64_t planes,
                                                        int H, int W, float rh, float rw)
{
    const int OH = 2 * H, OW = 2 * W;
    const int64_t total = planes * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ix = (int)(idx % W);
        const int64_t t = idx / W;
        const int iy = (int)(t % H);
        const int64_t pl = t / H;
        float wx[6];
        int ox0 = 2 * ix - 2;
#pragma unroll
        for (int e = 0; e < 6; ++e) {            // weight of output column ox0 + e on input column ix
            const int ox = ox0 + e;
            float wgt = 0.f;
            if (ox >= 0 && ox < OW) {
                const float w1r = rw * (float)ox;
                const int w1 = (int)w1r, w1p = (w1 < W - 1) ? 1 : 0;
                const float w1l = w1r - (float)w1;
                if (w1 == ix) wgt += 1.f - w1l;
                if (w1 + w1p == ix) wgt += w1l;   // (w1p = 0 at the last column: both terms land on it, as in the forward)
            }
            wx[e] = wgt;
        }
        float acc = 0.f;
        const int oy0 = 2 * iy - 2;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const int oy = oy0 + f;
            if (oy < 0 || oy >= OH) continue;
            const float h1r = rh * (float)oy;
            const int h1 = (int)h1r, h1p = (h1 < H - 1) ? 1 : 0;
            const float h1l = h1r - (float)h1;
            float wy = 0.f;
            if (h1 == iy) wy += 1.f - h1l;
            if (h1 + h1p == iy) wy += h1l;
            if (wy == 0.f) continue;
            const float *row = dout + (pl * OH + oy) * OW;
            float r = 0.f;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int ox = ox0 + e;
                if (wx[e] != 0.f) r += wx[e] * row[ox];
            }
            acc += wy * r;
        }
        din[idx] = acc;
    }
}
#pragma clang fp contract(fast)

int upsample2x_bwd(const float *dout, float *din, int64_t planes, int H, int W, hipStream_t stream)
{
    if (!dout || !din || planes <= 0 || H <= 0 || W <= 0) return VIT_EINVAL;
    const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const int64_t blocks = (planes * H * W + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_upsample2x_bwd, dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)), dim3(256), 0, stream, dout, din,
                       planes, H, W, rh, rw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int upsample2x_fwd(const float *in, float *out, int64_t planes, int H, int W, hipStream_t stream)
{
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || (W & 1)) return VIT_EINVAL;   // 2W % 4 == 0
    const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const int64_t total = planes * 2 * H * (2 * W / 4);
    const int64_t blocks = (total + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_upsample2x, dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)), dim3(256), 0, stream, in, out, planes,
                       H, W, rh, rw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
