// gsr_api.hip -- the C ABI declared in include/gsr.h (no torch, no exceptions).
#include <new>

#include "gsr_common.h"

namespace gsr {
int layout(const GsrDims &d, long long cap, GsrLayout &L);
int forward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *opac,
            const float *shs, long long cap, void *workspace, size_t workspace_bytes, float *image, float *depth,
            float *opacity, int32_t *radii, int32_t *n_touched, int32_t *status, hipStream_t stream);
int backward(const GsrDims &d, const GsrView *views, const float *means, const float *cov6, const float *shs,
             long long cap, void *workspace, size_t workspace_bytes, const float *dL_dimage, const float *dL_ddepth,
             float *dL_dmeans, float *dL_dcov6, float *dL_dopac, float *dL_dshs, float *dL_dmeans2D, float *dL_dtau,
             hipStream_t stream);
}  // namespace gsr

namespace gsr { thread_local hipError_t g_last_hip_error = hipSuccess; }

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
                        opacity, radii, n_touched, status, static_cast<hipStream_t>(stream));
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
                         dL_dmeans, dL_dcov6, dL_dopac, dL_dshs, dL_dmeans2D, dL_dtau, static_cast<hipStream_t>(stream));
}

__attribute__((visibility("default"))) GsrProfile *gsr_profile_create(int max_calls)
{
    if (max_calls <= 0) return nullptr;
    GsrProfile *p = new (std::nothrow) GsrProfile;
    if (!p) return nullptr;
    p->max_calls = max_calls; p->next_fwd = p->next_bwd = 0;
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

__attribute__((visibility("default"))) const char *gsr_version(void) { return "gsr-hip gfx950 0.1.0"; }

}  // extern "C"
