// vit_optim.hip -- the optimizer pass of the train step (model_wrapper_style.py:885-895, AdamW(lr, weight_decay=0.05,
// betas=(0.9, 0.95)) over ~600 parameter tensors / 1.05 G elements at full size) as ONE launch per parameter group.
//
// HBM-bound by construction: per element it reads p, g, m, v and writes p, m, v -- 28 bytes, 29.4 GB per step at full size,
// 4.7 ms at the ~6.3 TB/s a float4 copy reaches on MI355X.  The framework's multi-tensor kernel needs 64 launches / 8.9 ms for it
// (3.3 TB/s).  Layout: the tensors stay where they are (parameters in the module, gradients as views of the all-reduce buckets,
// moments in the optimizer state); a device table of chunks {p, g, m, v, step, n} tells each workgroup which <= 16 384 contiguous
// elements it owns.  Every lane moves float4s (the host marks a chunk scalar when one of its four pointers is not 16-byte aligned);
// eight independent 16-byte loads per lane are in flight before the first dependent instruction.
//
// Arithmetic = torch's fused AdamW kernel (aten/src/ATen/native/cuda/fused_adamw_impl.cu semantics, amsgrad = false, maximize = false):
//     g  = g / grad_scale                  (optional: the deferred clip coefficient, styl3r_amd/ddp.py clip_grad_norm_(defer_to=...))
//     p -= lr * wd * p ;  m = lerp(m, g, 1 - b1) ;  v = b2 v + (1 - b2) g g
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with t read from the parameter's own device-resident step counter (already incremented by the caller), the two bias corrections
// evaluated in double like the framework does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

namespace {
struct Coef { float lr_wd, b1, b2, eps, step_size, sqrt_bc2, scale; };

__device__ inline void adam1(float &p, float g, float &m, float &v, const Coef &c)
{
    g /= c.scale;
    p -= c.lr_wd * p;
    m = m + (1.f - c.b1) * (g - m);                  // lerp(m, g, 1 - b1)
    v = c.b2 * v + (1.f - c.b2) * g * g;
    const float denom = sqrtf(v) / c.sqrt_bc2 + c.eps;
    p -= c.step_size * m / denom;
}

__global__ void __launch_bounds__(256) k_adamw(const VitAdamChunk *__restrict__ chunks, float lr, float b1, float b2, float eps, float wd,
                                               const float *__restrict__ grad_scale)
{
    const VitAdamChunk ch = chunks[blockIdx.x];
    Coef c;
    {
        const double t = (double)*ch.step;
        const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
        c.lr_wd = lr * wd; c.b1 = b1; c.b2 = b2; c.eps = eps;
        c.step_size = (float)((double)lr / bc1);
        c.sqrt_bc2 = (float)sqrt(bc2);
        c.scale = grad_scale ? *grad_scale : 1.f;
    }
    const int n = ch.n, tid = threadIdx.x;
    uint32_t pm = 0;            // |max| of the updated parameter values of this lane (folded into *ch.amax for the f16x3 weight scale)
    auto seen = [&](float v) { pm = max(pm, __builtin_bit_cast(uint32_t, v) & 0x7fffffffu); };
    auto seen4 = [&](const float4 &v) { seen(v.x); seen(v.y); seen(v.z); seen(v.w); };
    if (ch.vec) {
        float4 *p4 = reinterpret_cast<float4 *>(ch.p), *m4 = reinterpret_cast<float4 *>(ch.m), *v4 = reinterpret_cast<float4 *>(ch.v);
        const float4 *g4 = reinterpret_cast<const float4 *>(ch.g);
        const int n4 = n >> 2;
        int i = tid;
        for (; i + 256 < n4; i += 512) {             // two float4 per operand in flight
            float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
            float4 P1 = p4[i + 256], G1 = g4[i + 256], M1 = m4[i + 256], V1 = v4[i + 256];
            adam1(P0.x, G0.x, M0.x, V0.x, c); adam1(P0.y, G0.y, M0.y, V0.y, c); adam1(P0.z, G0.z, M0.z, V0.z, c); adam1(P0.w, G0.w, M0.w, V0.w, c);
            adam1(P1.x, G1.x, M1.x, V1.x, c); adam1(P1.y, G1.y, M1.y, V1.y, c); adam1(P1.z, G1.z, M1.z, V1.z, c); adam1(P1.w, G1.w, M1.w, V1.w, c);
            p4[i] = P0; m4[i] = M0; v4[i] = V0;
            p4[i + 256] = P1; m4[i + 256] = M1; v4[i + 256] = V1;
            seen4(P0); seen4(P1);
        }
        for (; i < n4; i += 256) {
            float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
            adam1(P0.x, G0.x, M0.x, V0.x, c); adam1(P0.y, G0.y, M0.y, V0.y, c); adam1(P0.z, G0.z, M0.z, V0.z, c); adam1(P0.w, G0.w, M0.w, V0.w, c);
            p4[i] = P0; m4[i] = M0; v4[i] = V0;
            seen4(P0);
        }
        for (int j = (n4 << 2) + tid; j < n; j += 256) {
            float P = ch.p[j], M = ch.m[j], V = ch.v[j];
            adam1(P, ch.g[j], M, V, c);
            ch.p[j] = P; ch.m[j] = M; ch.v[j] = V;
            seen(P);
        }
    } else {
        for (int j = tid; j < n; j += 256) {
            float P = ch.p[j], M = ch.m[j], V = ch.v[j];
            adam1(P, ch.g[j], M, V, c);
            ch.p[j] = P; ch.m[j] = M; ch.v[j] = V;
            seen(P);
        }
    }
    if (ch.amax) {              // (workgroup-uniform) 64 slots, one per 128-byte line: csrc/vit_gemm_x6.hip amax_fold
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pm = max(pm, (uint32_t)__shfl_xor((int)pm, o, 64));
        uint32_t *w = ch.amax + ((blockIdx.x + (threadIdx.x >> 6)) & 63u) * 32;
        if ((threadIdx.x & 63) == 0 && pm > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, pm);
    }
}
}  // namespace

int adamw_step(const VitAdamChunk *chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
               const float *grad_scale, hipStream_t stream)
{
    if (n_chunks == 0) return VIT_OK;
    if (!chunks || n_chunks < 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return VIT_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_adamw, dim3(n_chunks), dim3(256), 0, stream, chunks, lr, beta1, beta2, eps, weight_decay, grad_scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
