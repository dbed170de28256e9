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
// ADD: out = upsample(in) + max(addend, 0) -- the 'gs' head's `feat_up(path_1) + input_merger(imgs)` (dpt_gs_head.py:146-148) with the
// input merger's ReLU applied while its pre-activation is read, in the same pass that writes the up-sampled map
template <bool ADD>
__device__ inline void upsample2x_quad(const float *__restrict__ in, const float *__restrict__ addend, float *__restrict__ out, int64_t pl, int oy,
                                       int ox4, int H, int W, float rh, float rw)
{
    const int OH = 2 * H, OW = 2 * W;
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
    if (ADD) {
        const float4 c = *reinterpret_cast<const float4 *>(addend + (pl * OH + oy) * OW + ox4 * 4);
        o.x += fmaxf(c.x, 0.f); o.y += fmaxf(c.y, 0.f); o.z += fmaxf(c.z, 0.f); o.w += fmaxf(c.w, 0.f);
    }
    *reinterpret_cast<float4 *>(out + (pl * OH + oy) * OW + ox4 * 4) = o;
}

template <bool ADD>
__global__ void __launch_bounds__(256) k_upsample2x(const float *__restrict__ in, const float *__restrict__ addend, float *__restrict__ out,
                                                    int64_t planes, int H, int W, float rh, float rw)
{
    const int OH = 2 * H, OW = 2 * W, OW4 = OW >> 2;
    const int64_t total = planes * OH * OW4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ox4 = (int)(idx % OW4);
        const int64_t t = idx / OW4;
        upsample2x_quad<ADD>(in, addend, out, t / OH, (int)(t % OH), ox4, H, W, rh, rw);
    }
}

// The same for output rows of a power-of-two number (<= 256) of 16-byte groups -- every map of the DPT heads -- and fewer than 2^31 output rows.
// A workgroup covers 256 / OW4 whole output rows.  The two input rows an output row interpolates between (2 W floats) are fetched by THAT row's
// OW4 = W / 2 threads with ONE coalesced 16-byte load each and parked in LDS (4 KiB per workgroup whatever W is); the sixteen values a thread
// needs then come out of LDS.  The generic kernel gathers them with sixteen 4-byte global loads per thread: it runs at 1.9 TB/s of algorithmic
// traffic where a copy of the same bytes runs at 5.3 (tools/probes/upsample_lab.py) -- the load instructions, not the bytes, are its limit.
// (row, group) is a shift and a mask, (plane, y) one 32-bit division.  Same arithmetic in the same order: bit-identical outputs.
template <bool ADD>
__global__ void __launch_bounds__(256) k_upsample2x_p2(const float *__restrict__ in, const float *__restrict__ addend, float *__restrict__ out,
                                                       uint32_t rows, int H, int W, float rh, float rw, int log2_ow4)
{
    __shared__ float4 s_rows[256];                       // [row of the workgroup][input row 0 / 1][W floats] = 256 x 16 bytes
    const uint32_t rloc = threadIdx.x >> log2_ow4, c = threadIdx.x & ((1u << log2_ow4) - 1u);
    const uint32_t row = min((blockIdx.x << (8 - log2_ow4)) + rloc, rows - 1u);      // (clamped, not returned: the barrier below is for everyone)
    const bool live = (blockIdx.x << (8 - log2_ow4)) + rloc < rows;
    const uint32_t OH = 2u * (uint32_t)H, pl = row / OH;
    const int oy = (int)(row - pl * OH), OW = 2 * W;
    const float h1r = rh * (float)oy;
    const int h1 = (int)h1r, h1p = (h1 < H - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    {   // thread c of the row: 16-byte group c of the 2 W floats [row h1 | row h1 + h1p]
        const int w4 = W >> 2, which = (int)c >= w4 ? 1 : 0, g4 = (int)c - which * w4;
        const float *src = in + ((int64_t)pl * H + h1 + which * h1p) * W + 4 * g4;
        s_rows[threadIdx.x] = *reinterpret_cast<const float4 *>(src);
    }
    __syncthreads();
    if (!live) return;
    const float *r0 = reinterpret_cast<const float *>(s_rows + (rloc << log2_ow4)), *r1 = r0 + W;
    float4 o;
    float *po = reinterpret_cast<float *>(&o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ox = (int)c * 4 + e;
        const float w1r = rw * (float)ox;
        const int w1 = (int)w1r, w1p = (w1 < W - 1) ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
        po[e] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
    }
    const int64_t obase = ((int64_t)pl * OH + oy) * OW + (int64_t)c * 4;
    if (ADD) {
        const float4 a4 = *reinterpret_cast<const float4 *>(addend + obase);
        o.x += fmaxf(a4.x, 0.f); o.y += fmaxf(a4.y, 0.f); o.z += fmaxf(a4.z, 0.f); o.w += fmaxf(a4.w, 0.f);
    }
    *reinterpret_cast<float4 *>(out + obase) = o;
}

// 7x7 / stride 1 / padding 3 patches of a 3-channel image as 160 "channels" (147 = 3 x 7 x 7 taps in the weight's (ci, ky, kx)
// order + 13 zero channels: the bf16x6 convolution kernels contract over multiples of 16): the `input_merger` convolution
// Conv2d(3, 256, 7, 1, 3) (dpt_gs_head.py:113-118) then IS a 1x1 convolution over these planes on vit_conv_x6_fwd / _wgrad.
__global__ void __launch_bounds__(256) k_im2col7(const float *__restrict__ img, float *__restrict__ cols, int B, int H, int W)
{
    const int W4 = W >> 2;
    const int64_t total = (int64_t)B * 160 * H * W4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(idx % W4);
        int64_t t = idx / W4;
        const int y = (int)(t % H); t /= H;
        const int k = (int)(t % 160);
        const int64_t b = t / 160;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < 147) {
            const int ci = k / 49, r = k - ci * 49, ky = r / 7, kx = r - ky * 7;
            const int ys = y + ky - 3;
            if (ys >= 0 && ys < H) {
                const float *row = img + ((b * 3 + ci) * H + ys) * (int64_t)W;
                float *po = reinterpret_cast<float *>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xs = x4 * 4 + e + kx - 3;
                    po[e] = (xs >= 0 && xs < W) ? row[xs] : 0.f;
                }
            }
        }
        *reinterpret_cast<float4 *>(cols + ((b * 160 + k) * H + y) * (int64_t)W + x4 * 4) = o;
    }
}
// Backward as a GATHER (the framework scatters with atomics): input pixel (iy, ix) collects w_y * w_x * dout from the
// output rows / columns whose source cell touches it.  With scale (H-1)/(2H-1) < 1/2 those are among the six candidates
// 2i-2 .. 2i+3; each candidate's cell index is recomputed with the forward's own float arithmetic, so forward and
// backward agree on every floor() decision.  One pass: dout read once (cached across neighbours), din written once.
// `rows0`: the address of output row `oy_base` of the pixel's plane (global memory, or the workgroup's LDS copy of a band of rows)
__device__ inline float upsample2x_bwd_pixel(const float *__restrict__ rows0, int oy_base, int iy, int ix, int H, int W, float rh, float rw)
{
    const int OH = 2 * H, OW = 2 * W;
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
        const float *row = rows0 + (int64_t)(oy - oy_base) * OW;
        float r = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) {    // branch-free: columns outside the row read a clamped address and are SELECTED away
            const float t = wx[e] * row[min(max(ox0 + e, 0), OW - 1)];   // (v_cndmask, not a multiply by 0: 0 * Inf of an overflowed border gradient would be NaN)
            r += wx[e] != 0.f ? t : 0.f;
        }
        acc += wy * r;
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_upsample2x_bwd(const float *__restrict__ dout, float *__restrict__ din, int64_t planes,
                                                        int H, int W, float rh, float rw)
{
    const int64_t total = planes * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ix = (int)(idx % W);
        const int64_t t = idx / W;
        din[idx] = upsample2x_bwd_pixel(dout + (t / H) * (int64_t)(4 * H) * W, 0, (int)(t % H), ix, H, W, rh, rw);
    }
}

// power-of-two width, fewer than 2^31 input pixels: shift / mask / one 32-bit division instead of two 64-bit divisions (see k_upsample2x_p2)
__global__ void __launch_bounds__(256) k_upsample2x_bwd_p2(const float *__restrict__ dout, float *__restrict__ din, uint32_t total, int H, int W,
                                                           float rh, float rw, int log2_w)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t t = idx >> log2_w, pl = t / (uint32_t)H;
    din[idx] = upsample2x_bwd_pixel(dout + (int64_t)pl * (4 * H) * W, 0, (int)(t - pl * (uint32_t)H), (int)(idx & ((1u << log2_w) - 1u)), H, W, rh, rw);
}

// The backward through LDS (power-of-two W <= 256 and H, at least 256 pixels per plane): a workgroup owns R = 256 / W input rows of one plane; the
// 2 R + 4 output rows they collect from are copied to LDS with coalesced 16-byte loads, and the 36 candidate taps of a pixel come out of LDS
// instead of 36 four-byte global loads per thread (the forward's finding, k_upsample2x_p2).  Same taps, same order: bit-identical gradients.
__global__ void __launch_bounds__(256) k_upsample2x_bwd_lds(const float *__restrict__ dout, float *__restrict__ din, int H, int W, float rh, float rw,
                                                            int log2_w)
{
    extern __shared__ float4 s_band4[];
    const int R = 256 >> log2_w, OH = 2 * H, OW = 2 * W;
    const int bands = H / R;                                   // workgroups per plane
    const uint32_t pl = blockIdx.x / (uint32_t)bands;
    const int iy0 = (int)(blockIdx.x - pl * (uint32_t)bands) * R;
    const int oy_lo = max(2 * iy0 - 2, 0), oy_hi = min(2 * (iy0 + R - 1) + 3, OH - 1);
    const float *plane = dout + (int64_t)pl * OH * OW;
    const int n4 = (oy_hi - oy_lo + 1) * (OW >> 2);
    const float4 *src = reinterpret_cast<const float4 *>(plane + (int64_t)oy_lo * OW);
    for (int i = threadIdx.x; i < n4; i += 256) s_band4[i] = src[i];
    __syncthreads();
    const int ix = (int)(threadIdx.x & ((1u << log2_w) - 1u)), iy = iy0 + (int)(threadIdx.x >> log2_w);
    din[((int64_t)pl * H + iy) * W + ix] = upsample2x_bwd_pixel(reinterpret_cast<const float *>(s_band4), oy_lo, iy, ix, H, W, rh, rw);
}
#pragma clang fp contract(fast)

// the fast forward kernel's precondition: W a multiple of 4, 2W / 4 a power of two <= 256, fewer than 2^31 output rows
//  ... and a 16-byte aligned INPUT: the fast kernels read `in` / `dout` with float4 loads (the generic ones only ever needed the
// output and the addend aligned)
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool upsample_p2_ok(int64_t planes, int H, int W, const void *in)
{
    const int ow4 = 2 * W / 4;
    return (W & 3) == 0 && ow4 <= 256 && (ow4 & (ow4 - 1)) == 0 && planes * 2 * H < (int64_t)0x7fffffff && aligned16(in);
}

int upsample2x_bwd(const float *dout, float *din, int64_t planes, int H, int W, hipStream_t stream)
{
    if (!dout || !din || planes <= 0 || H <= 0 || W <= 0) return VIT_EINVAL;
    const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const int64_t blocks = (planes * H * W + 255) / 256;
    (void)hipGetLastError();
    const int R = (W & (W - 1)) == 0 && W <= 256 && W >= 2 ? 256 / W : 0;
    if (R && (H & (H - 1)) == 0 && H >= R && planes * (H / R) < (int64_t)0x7fffffff && aligned16(dout))
        hipLaunchKernelGGL(k_upsample2x_bwd_lds, dim3((unsigned)(planes * (H / R))), dim3(256), (size_t)(2 * R + 4) * 2 * W * sizeof(float), stream, dout,
                           din, H, W, rh, rw, __builtin_ctz((unsigned)W));
    else if ((W & (W - 1)) == 0 && planes * H * W < (int64_t)0x7fffffff)
        hipLaunchKernelGGL(k_upsample2x_bwd_p2, dim3((unsigned)blocks), dim3(256), 0, stream, dout, din, (uint32_t)(planes * H * W), H, W, rh, rw,
                           __builtin_ctz((unsigned)W));
    else
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
    if (upsample_p2_ok(planes, H, W, in)) {
        const int l2 = __builtin_ctz((unsigned)(2 * W / 4));
        const int64_t rows = planes * 2 * H;
        hipLaunchKernelGGL(k_upsample2x_p2<false>, dim3((unsigned)((rows + (256 >> l2) - 1) >> (8 - l2))), dim3(256), 0, stream, in,
                           (const float *)nullptr, out, (uint32_t)rows, H, W, rh, rw, l2);
    } else
        hipLaunchKernelGGL(k_upsample2x<false>, dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)), dim3(256), 0, stream, in,
                           (const float *)nullptr, out, planes, H, W, rh, rw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

// 3x3 / stride 1 / padding 1 patches of an NCHW tensor as pixel-major rows: cols[(b, y, x)][(tap, ci)] = f(in[b, ci, y + dy - 1, x + dx - 1])
// (zero outside the image; f = ReLU when `relu`), tap = 3 dy + dx.  With it the weight gradient of a 3x3 convolution over FEW pixels
// (the 8 x 8 .. 64 x 64 stages of the DPT heads) is the Linear weight gradient dW (Co, 9 Ci) = dY^T . cols on vit_linear_x6_wgrad.
// One workgroup = 32 pixels of one image row x 32 channels: the three source rows go through LDS (coalesced along x on the way in,
// along ci on the way out; row stride 35 floats: conflict-free for lanes that differ in the channel).
__global__ void __launch_bounds__(256) k_im2col3_rows(const float *__restrict__ in, float *__restrict__ cols, int B, int Ci, int H, int W, int relu)
{
    __shared__ float s[3][32][35];
    const int x0 = blockIdx.x * 32, y = blockIdx.y;
    const int ctiles = (Ci + 31) / 32;
    const int b = blockIdx.z / ctiles, c0 = (blockIdx.z % ctiles) * 32;
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 32 * 34; i += 256) {
        const int px = i % 34, ch = (i / 34) % 32, r = i / (34 * 32);
        const int ys = y + r - 1, xs = x0 + px - 1, c = c0 + ch;
        float v = 0.f;
        if (ys >= 0 && ys < H && xs >= 0 && xs < W && c < Ci) v = in[(((int64_t)b * Ci + c) * H + ys) * W + xs];
        s[r][ch][px] = relu ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
    const int ch = tid & 31, pr = tid >> 5;
    if (c0 + ch >= Ci) return;
    for (int px = pr; px < 32; px += 8) {
        const int x = x0 + px;
        if (x >= W) break;
        float *row = cols + (((int64_t)b * H + y) * W + x) * (9 * (int64_t)Ci) + c0 + ch;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) row[(int64_t)tap * Ci] = s[tap / 3][ch][px + tap % 3];
    }
}

int im2col3_rows(const float *in, float *cols, int B, int Ci, int H, int W, int relu, hipStream_t stream)
{
    if (!in || !cols || B <= 0 || Ci <= 0 || H <= 0 || W <= 0) return VIT_EINVAL;
    const int64_t gz = (int64_t)B * ((Ci + 31) / 32);
    if (gz > 65535 || H > 65535) return VIT_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_im2col3_rows, dim3((W + 31) / 32, H, (unsigned)gz), dim3(256), 0, stream, in, cols, B, Ci, H, W, relu);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int upsample2x_add_relu_fwd(const float *in, const float *addend, float *out, int64_t planes, int H, int W, hipStream_t stream)
{
    if (!in || !addend || !out || planes <= 0 || H <= 0 || W <= 0 || (W & 1)) return VIT_EINVAL;
    const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const int64_t total = planes * 2 * H * (2 * W / 4);
    const int64_t blocks = (total + 255) / 256;
    (void)hipGetLastError();
    if (upsample_p2_ok(planes, H, W, in)) {
        const int l2 = __builtin_ctz((unsigned)(2 * W / 4));
        const int64_t rows = planes * 2 * H;
        hipLaunchKernelGGL(k_upsample2x_p2<true>, dim3((unsigned)((rows + (256 >> l2) - 1) >> (8 - l2))), dim3(256), 0, stream, in, addend,
                           out, (uint32_t)rows, H, W, rh, rw, l2);
    } else
        hipLaunchKernelGGL(k_upsample2x<true>, dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)), dim3(256), 0, stream, in, addend,
                           out, planes, H, W, rh, rw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int im2col7(const float *img, float *cols, int B, int H, int W, hipStream_t stream)
{
    if (!img || !cols || B <= 0 || H <= 0 || W <= 0 || (W & 3)) return VIT_EINVAL;
    const int64_t total = (int64_t)B * 160 * H * (W >> 2);
    const int64_t blocks = (total + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_im2col7, dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)), dim3(256), 0, stream, img, cols, B, H, W);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
// ---- ReLU -> Dropout of the 'gs_params' DPT heads (dpt_block.py:332-340: ReLU(True), Dropout(0.1)) on their 256^2 x 256-channel
// tensors (1.3 GB at 20 views) as ONE pass each way.  Forward: y = keep(i) ? max(x, 0) / (1 - p) : 0 with keep drawn by a
// counter-based generator (Philox-4x32-10 keyed by (seed, element index / 4): no mask tensor is written); y > 0 exactly where
// the gradient passes, so the backward needs y only: dx = y > 0 ? g / (1 - p) : 0.  The framework runs clamp + fused_dropout
// forward and masked_scale + threshold_backward backward, two 2.7 GB passes each way plus the mask.
__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key)
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

__global__ void __launch_bounds__(256) k_relu_dropout_fwd(const float *__restrict__ x, float *__restrict__ y, int64_t n4, uint32_t thresh,
                                                          float scale, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), 0u, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        float4 o;                                  // keep with probability 1 - p: a uniform 32-bit draw below thresh = (1 - p) 2^32
        o.x = (r.x < thresh && v.x > 0.f) ? v.x * scale : 0.f;
        o.y = (r.y < thresh && v.y > 0.f) ? v.y * scale : 0.f;
        o.z = (r.z < thresh && v.z > 0.f) ? v.z * scale : 0.f;
        o.w = (r.w < thresh && v.w > 0.f) ? v.w * scale : 0.f;
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

__global__ void __launch_bounds__(256) k_relu_dropout_bwd(const float *__restrict__ y, const float *__restrict__ g, float *__restrict__ dx,
                                                          int64_t n4, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4 *>(y)[i], d = reinterpret_cast<const float4 *>(g)[i];
        reinterpret_cast<float4 *>(dx)[i] = make_float4(a.x > 0.f ? d.x * scale : 0.f, a.y > 0.f ? d.y * scale : 0.f,
                                                         a.z > 0.f ? d.z * scale : 0.f, a.w > 0.f ? d.w * scale : 0.f);
    }
}

int relu_dropout_fwd(const float *x, float *y, int64_t n, float p, uint64_t seed, hipStream_t stream)
{
    if (!x || !y || n <= 0 || (n & 3) || !(p >= 0.f && p < 1.f)) return VIT_EINVAL;
    const double keep = 1.0 - (double)p;
    const uint32_t thresh = keep >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(keep * 4294967296.0);
    (void)hipGetLastError();
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(k_relu_dropout_fwd, dim3((unsigned)((n4 + 255) / 256 < 256 * 64 ? (n4 + 255) / 256 : 256 * 64)), dim3(256), 0, stream, x, y, n4, thresh,
                       (float)(1.0 / keep), seed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int relu_dropout_bwd(const float *y, const float *g, float *dx, int64_t n, float p, hipStream_t stream)
{
    if (!y || !g || !dx || n <= 0 || (n & 3) || !(p >= 0.f && p < 1.f)) return VIT_EINVAL;
    (void)hipGetLastError();
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(k_relu_dropout_bwd, dim3((unsigned)((n4 + 255) / 256 < 256 * 64 ? (n4 + 255) / 256 : 256 * 64)), dim3(256), 0, stream, y, g, dx, n4,
                       (float)(1.0 / (1.0 - (double)p)));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
