// gsr_api.hip -- the C ABI declared in include/gsr.h (no torch, no exceptions).
#include <new>

#include "gsr_common.h"

namespace gsr {
int layout(const GsrDims &d, long long cap, GsrLayout &L);
int forward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *opac,
            const float *shs, long long cap, void *workspace, size_t workspace_bytes, float *image, float *depth,
            float *opacity, int32_t *radii, int32_t *n_touched, int32_t *status, const GsrFused *fx, hipStream_t stream);
int backward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *shs,
             long long cap, void *workspace, size_t workspace_bytes, const float *dL_dimage, const float *dL_ddepth,
             float *dL_dmeans, float *dL_dcov6, float *dL_dopac, float *dL_dshs, float *dL_dmeans2D, float *dL_dtau,
             const GsrFused *fx, hipStream_t stream);
}  // namespace gsr

namespace gsr {
thread_local hipError_t g_last_hip_error = hipSuccess;

// ---- view set-up kernel: one thread per view ---------------------------------------------
#pragma clang fp contract(off)
__device__ inline void inverse4(const float *m, float *inv)
{
    // Gauss-Jordan with partial pivoting on [m | I] (row-major)
    float a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = (r == c) ? 1.f : 0.f; }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > col && fabsf(a[r][col]) > best) { best = fabsf(a[r][col]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && piv != col) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { float t = a[col][c]; a[col][c] = a[r][c]; a[r][c] = t; }
            }
        const float d = 1.0f / a[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const float f = a[r][col];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) inv[4 * r + c] = a[r][4 + c];
}

__device__ inline void ray_dir(const float *Ki, float x, float y, float *d)
{
    // K^-1 [x, y, 1], normalised
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = Ki[3 * i] * x + Ki[3 * i + 1] * y + Ki[3 * i + 2];
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    d[0] = v[0] / n; d[1] = v[1] / n; d[2] = v[2] / n;
}

__global__ void k_build_views(const float *__restrict__ c2w, const float *__restrict__ K, const float *__restrict__ near,
                              const float *__restrict__ far, const float *__restrict__ bg, int V, int scale_invariant,
                              GsrView *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float E[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) E[i] = c2w[16 * v + i];
    float nr = near[v], fr = far[v], scale = 1.f;
    if (scale_invariant) {            // cuda_splatting.py:65-72
        scale = 1.f / nr;
        E[3] *= scale; E[7] *= scale; E[11] *= scale;
        nr = nr * scale; fr = fr * scale;
    }
    // intrinsics inverse (3x3, adjugate / determinant)
    const float *k = K + 9 * v;
    float Ki[9];
    {
        const float c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
        const float det = k[0] * c00 + k[1] * c01 + k[2] * c02, id = 1.0f / det;
        Ki[0] = c00 * id; Ki[1] = (k[2] * k[7] - k[1] * k[8]) * id; Ki[2] = (k[1] * k[5] - k[2] * k[4]) * id;
        Ki[3] = c01 * id; Ki[4] = (k[0] * k[8] - k[2] * k[6]) * id; Ki[5] = (k[2] * k[3] - k[0] * k[5]) * id;
        Ki[6] = c02 * id; Ki[7] = (k[1] * k[6] - k[0] * k[7]) * id; Ki[8] = (k[0] * k[4] - k[1] * k[3]) * id;
    }
    float l[3], r[3], t[3], b[3];
    ray_dir(Ki, 0.f, 0.5f, l); ray_dir(Ki, 1.f, 0.5f, r); ray_dir(Ki, 0.5f, 0.f, t); ray_dir(Ki, 0.5f, 1.f, b);
    const float fovx = acosf(l[0] * r[0] + l[1] * r[1] + l[2] * r[2]);
    const float fovy = acosf(t[0] * b[0] + t[1] * b[1] + t[2] * b[2]);
    const float tx = tanf(0.5f * fovx), ty = tanf(0.5f * fovy);
    // projection (column-vector form P), stored transposed
    const float top = ty * nr, right = tx * nr;
    float P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = 0.f;
    P[0] = 2.f * nr / (right - (-right));
    P[5] = 2.f * nr / (top - (-top));
    P[2] = (right + (-right)) / (right - (-right));
    P[6] = (top + (-top)) / (top - (-top));
    P[14] = 1.f;
    P[10] = fr / (fr - nr);
    P[11] = -(fr * nr) / (fr - nr);
    float Ei[16];
    inverse4(E, Ei);
    GsrView &o = out[v];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            o.viewmatrix[4 * rr + cc] = Ei[4 * cc + rr];          // inverse(c2w)^T
            o.projmatrix_raw[4 * rr + cc] = P[4 * cc + rr];        // P^T
        }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            float acc = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc += o.viewmatrix[4 * rr + kk] * o.projmatrix_raw[4 * kk + cc];
            o.projmatrix[4 * rr + cc] = acc;
        }
    o.campos[0] = E[3]; o.campos[1] = E[7]; o.campos[2] = E[11];
    o.tanfovx = tx; o.tanfovy = ty;
    o.bg[0] = bg[3 * v]; o.bg[1] = bg[3 * v + 1]; o.bg[2] = bg[3 * v + 2];
    o.scale = scale;
#pragma unroll
    for (int i = 0; i < 7; ++i) o.pad[i] = 0.f;
}
#pragma clang fp contract(fast)
}  // namespace gsr

extern "C" {

__attribute__((visibility("default"))) const char *gsr_last_error(void)
{
    return gsr::g_last_hip_error == hipSuccess ? "" : hipGetErrorString(gsr::g_last_hip_error);
}

__attribute__((visibility("default"))) int gsr_workspace_layout(const GsrDims *dims, int64_t pair_capacity, GsrLayout *out)
{
    if (!dims || !out) return GSR_EINVAL;
    return gsr::layout(*dims, pair_capacity, *out);
}

__attribute__((visibility("default"))) int gsr_forward(const GsrDims *dims, const GsrView *views, const float *means,
                                                       const float *cov6, const float *opac, const float *shs,
                                                       int64_t pair_capacity, void *workspace, size_t workspace_bytes,
                                                       float *image, float *depth, float *opacity, int32_t *radii,
                                                       int32_t *n_touched, int32_t *status, void *stream)
{
    if (!dims) return GSR_EINVAL;
    return gsr::forward(*dims, views, means, cov6, opac, shs, pair_capacity, workspace, workspace_bytes, image, depth,
                        opacity, radii, n_touched, status, nullptr, static_cast<hipStream_t>(stream));
}

__attribute__((visibility("default"))) int gsr_forward_fused(const GsrDims *dims, const GsrView *views, const float *means,
                                                             const float *cov6, const float *opac, const float *shs,
                                                             int64_t pair_capacity, void *workspace, size_t workspace_bytes,
                                                             float *image, float *depth, float *opacity, int32_t *radii,
                                                             int32_t *n_touched, int32_t *status, const GsrFused *fx, void *stream)
{
    if (!dims) return GSR_EINVAL;
    return gsr::forward(*dims, views, means, cov6, opac, shs, pair_capacity, workspace, workspace_bytes, image, depth,
                        opacity, radii, n_touched, status, fx, static_cast<hipStream_t>(stream));
}

__attribute__((visibility("default"))) int gsr_backward(const GsrDims *dims, const GsrView *views, const float *means,
                                                        const float *cov6, const float *shs, int64_t pair_capacity,
                                                        void *workspace, size_t workspace_bytes, const float *dL_dimage,
                                                        const float *dL_ddepth, float *dL_dmeans, float *dL_dcov6,
                                                        float *dL_dopac, float *dL_dshs, float *dL_dmeans2D,
                                                        float *dL_dtau, void *stream)
{
    if (!dims) return GSR_EINVAL;
    return gsr::backward(*dims, views, means, cov6, shs, pair_capacity, workspace, workspace_bytes, dL_dimage, dL_ddepth,
                         dL_dmeans, dL_dcov6, dL_dopac, dL_dshs, dL_dmeans2D, dL_dtau, nullptr, static_cast<hipStream_t>(stream));
}

__attribute__((visibility("default"))) int gsr_backward_fused(const GsrDims *dims, const GsrView *views, const float *means,
                                                              const float *cov6, const float *shs, int64_t pair_capacity,
                                                              void *workspace, size_t workspace_bytes, const float *dL_dimage,
                                                              const float *dL_ddepth, float *dL_dmeans, float *dL_dcov6,
                                                              float *dL_dopac, float *dL_dshs, float *dL_dmeans2D,
                                                              float *dL_dtau, const GsrFused *fx, void *stream)
{
    if (!dims) return GSR_EINVAL;
    return gsr::backward(*dims, views, means, cov6, shs, pair_capacity, workspace, workspace_bytes, dL_dimage, dL_ddepth,
                         dL_dmeans, dL_dcov6, dL_dopac, dL_dshs, dL_dmeans2D, dL_dtau, fx, static_cast<hipStream_t>(stream));
}

__attribute__((visibility("default"))) int gsr_build_views(const float *c2w, const float *K, const float *near,
                                                           const float *far, const float *bg, int32_t V,
                                                           int32_t scale_invariant, GsrView *out, void *stream)
{
    if (!c2w || !K || !near || !far || !bg || !out || V <= 0) return GSR_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(gsr::k_build_views, dim3((V + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), c2w, K, near,
                       far, bg, V, scale_invariant, out);
    return gsr::launch_status();
}

__attribute__((visibility("default"))) GsrProfile *gsr_profile_create(int max_calls)
{
    if (max_calls <= 0) return nullptr;
    GsrProfile *p = new (std::nothrow) GsrProfile;
    if (!p) return nullptr;
    p->max_calls = max_calls; p->next_fwd = p->next_bwd = 0; p->mask = (1u << GSR_N_STAGES) - 1u;
    const size_t n = (size_t)max_calls * GSR_N_STAGES * 2;
    p->ev = new (std::nothrow) hipEvent_t[n];
    if (!p->ev) { delete p; return nullptr; }
    for (size_t i = 0; i < n; ++i)
        if (hipEventCreate(&p->ev[i]) != hipSuccess) { p->ev[i] = nullptr; }
    return p;
}

__attribute__((visibility("default"))) void gsr_profile_destroy(GsrProfile *p)
{
    if (!p) return;
    const size_t n = (size_t)p->max_calls * GSR_N_STAGES * 2;
    for (size_t i = 0; i < n; ++i)
        if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev;
    delete p;
}

__attribute__((visibility("default"))) int gsr_profile_read(GsrProfile *p, float *ms_sum, int32_t *count)
{
    if (!p || !ms_sum || !count) return GSR_EINVAL;
    for (int s = 0; s < GSR_N_STAGES; ++s) { ms_sum[s] = 0.f; count[s] = 0; }
    for (int s = 0; s < GSR_N_STAGES; ++s) {
        if (!((p->mask >> s) & 1u)) continue;
        const int n = (s <= GSR_STAGE_COMPOSITE_FWD) ? p->next_fwd : p->next_bwd;
        for (int c = 0; c < n; ++c) {
            float ms = 0.f;
            if (hipEventSynchronize(p->at(c, s, 1)) != hipSuccess) return GSR_ELAUNCH;
            if (hipEventElapsedTime(&ms, p->at(c, s, 0), p->at(c, s, 1)) != hipSuccess) return GSR_ELAUNCH;
            ms_sum[s] += ms; count[s] += 1;
        }
    }
    p->next_fwd = p->next_bwd = 0;
    return GSR_OK;
}

__attribute__((visibility("default"))) int gsr_profile_set_stages(GsrProfile *p, uint32_t stage_mask)
{
    if (!p || p->next_fwd || p->next_bwd) return GSR_EINVAL;      // between reads only: a half-recorded profile cannot change its stages
    p->mask = stage_mask & ((1u << GSR_N_STAGES) - 1u);
    return GSR_OK;
}

__attribute__((visibility("default"))) const char *gsr_version(void) { return "gsr-hip gfx950 0.6.0"; }

}  // extern "C"
