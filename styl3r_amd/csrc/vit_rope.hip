// vit_rope.hip -- fused in-place 2-D RoPE for gfx950 (replaces the curope CUDA extension).
//
// Pure bandwidth: one read and one write of the token buffer.  A lane owns one rotation PAIR
// group: the four floats (u_Y[d], v_Y[d], u_X[d], v_X[d]) of one (token, head, d); consecutive lanes
// take consecutive d, so every quarter is read and written as a coalesced run of D/4 floats and the
// 64 lanes of a wavefront cover 64/(D/4) heads of a token (D = 64 -> 4 heads per wavefront).
// cos/sin come from a (P, D/4) table (positions are small integers), not from per-element
// sinf/cosf/powf, so the kernel stays memory-bound.  Contraction is off and the operation order is
// the reference fallback's ( t*cos + rotate_half(t)*sin, pos_embed.py:135-139 ) => bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) k_rope2d(float *__restrict__ tokens, const int64_t *__restrict__ pos,
                                                const float *__restrict__ cos_tab, const float *__restrict__ sin_tab,
                                                int N, int H, int Q, int P, long long total, int64_t sb, int64_t sn,
                                                int64_t sh, float sign)
{
    // flat index over (b, n, h, d<Q), d fastest
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % Q);
        long long r = i / Q;
        const int h = (int)(r % H); r /= H;
        const int n = (int)(r % N);
        const long long b = r / N;
        const int64_t py = pos[(b * N + n) * 2 + 0], px = pos[(b * N + n) * 2 + 1];
        float *t = tokens + b * sb + n * sn + h * sh;
        const float cy = cos_tab[py * Q + d], sy = sign * sin_tab[py * Q + d];
        const float cx = cos_tab[px * Q + d], sx = sign * sin_tab[px * Q + d];
        const float uy = t[d], vy = t[Q + d], ux = t[2 * Q + d], vx = t[3 * Q + d];
        t[d] = uy * cy + (-vy) * sy;
        t[Q + d] = vy * cy + uy * sy;
        t[2 * Q + d] = ux * cx + (-vx) * sx;
        t[3 * Q + d] = vx * cx + ux * sx;
    }
}
#pragma clang fp contract(fast)

int rope2d(float *tokens, const int64_t *positions, const float *cos_tab, const float *sin_tab, int B, int N, int H,
           int D, int P, int64_t sb, int64_t sn, int64_t sh, float sign, hipStream_t stream)
{
    if (!tokens || !positions || !cos_tab || !sin_tab) return VIT_EINVAL;
    if (B <= 0 || N <= 0 || H <= 0 || D <= 0 || (D & 3) || P <= 0) return VIT_EINVAL;
    const int Q = D / 4;
    const long long total = (long long)B * N * H * Q;
    const int blocks = (int)((total + 255) / 256 < 256 * 8 ? (total + 255) / 256 : 256 * 8);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_rope2d, dim3(blocks), dim3(256), 0, stream, tokens, positions, cos_tab, sin_tab, N, H, Q, P, total,
                       sb, sn, sh, sign);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
