// gsr_loss.hip -- the MSE consumer of the rendered colour (src/loss/loss_mse.py:22-31:
// weight * ((prediction.color - target) ** 2).mean()) as two HBM-bound kernels instead of the seven
// elementwise / reduce launches of the torch expression.
//
// forward : one pass over both images, per-workgroup partial sums, the LAST workgroup to finish (ticket
//           counter) adds the partials in index order -> deterministic, one launch.
// backward: dL/dpred = (2 * weight / n) * g * (pred - target), one pass; g is read from device memory
//           (the upstream scalar gradient), so nothing syncs with the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"
#include "gsr_common.h"

namespace gsr {

constexpr int MSE_BLOCK = 256, MSE_MAX_GROUPS = 512, MSE_FWD_GROUPS = 256;
// Forward: few, fat workgroups.  Every workgroup ends with a device-scope release (__threadfence: an L2 write-back on this multi-die part) and a
// ticket; measured on the headline image batch (7.9 M floats): 2 048 workgroups 108 us, 512 (one ticket for all) 24.7, 256 19.4, 128 20.4.
// Tickets: one per cluster of MSE_CLUSTER workgroups, each on its own 128-byte line, and one for the clusters.
constexpr int MSE_CLUSTER = 32, MSE_TICKET_STRIDE = 32, MSE_MAX_CLUSTERS = MSE_MAX_GROUPS / MSE_CLUSTER;

__device__ inline float block_sum_256(float v, float *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];          // same value in every thread
}

__global__ void __launch_bounds__(MSE_BLOCK) k_mse_fwd(const float *__restrict__ pred, const float *__restrict__ target,
                                                       long long n, float weight, float *__restrict__ scratch,
                                                       float *__restrict__ out)
{
    __shared__ float sh[4];
    __shared__ unsigned last;
    const long long n4 = n >> 2;
    const float4 *p4 = reinterpret_cast<const float4 *>(pred), *t4 = reinterpret_cast<const float4 *>(target);
    float acc = 0.f;
    // (the loop keeps four 16-byte loads per operand in flight)
    const long long stride = (long long)gridDim.x * MSE_BLOCK;
    long long i = (long long)blockIdx.x * MSE_BLOCK + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = p4[i + u * stride]; b[u] = t4[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float dx = a[u].x - b[u].x, dy = a[u].y - b[u].y, dz = a[u].z - b[u].z, dw = a[u].w - b[u].w;
            acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    for (; i < n4; i += stride) {
        const float4 a = p4[i], b = t4[i];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;   // tail (n % 4 elements)
        if (i < n) { const float d = pred[i] - target[i]; acc += d * d; }
    }
    const float s = block_sum_256(acc, sh);
    unsigned *tickets = reinterpret_cast<unsigned *>(scratch + MSE_MAX_GROUPS);      // [cluster] at stride MSE_TICKET_STRIDE, then the clusters' ticket
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = s;
        __threadfence();
        const unsigned cl = blockIdx.x / MSE_CLUSTER, ncl = (gridDim.x + MSE_CLUSTER - 1) / MSE_CLUSTER;
        const unsigned members = min((unsigned)MSE_CLUSTER, gridDim.x - cl * MSE_CLUSTER);
        unsigned *t1 = tickets + cl * MSE_TICKET_STRIDE, *t2 = tickets + MSE_MAX_CLUSTERS * MSE_TICKET_STRIDE;
        unsigned l = 0u;
        if (atomicAdd(t1, 1u) == members - 1u) {        // last of its cluster: re-arm the cluster's ticket, take the second level
            *t1 = 0u;
            __threadfence();
            if (atomicAdd(t2, 1u) == ncl - 1u) { *t2 = 0u; l = 1u; }
        }
        last = l;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += MSE_BLOCK) t += __builtin_nontemporal_load(scratch + i);
    __syncthreads();                                    // sh reuse
    t = block_sum_256(t, sh);
    if (threadIdx.x == 0) out[0] = weight * (t / (float)n);      // (the tickets were re-armed by their last arrivers)
}

__global__ void __launch_bounds__(MSE_BLOCK) k_mse_bwd(const float *__restrict__ pred, const float *__restrict__ target,
                                                       const float *__restrict__ g, long long n, float weight,
                                                       float *__restrict__ grad)
{
    const float c = 2.0f * weight / (float)n * g[0];
    const long long n4 = n >> 2;
    const float4 *p4 = reinterpret_cast<const float4 *>(pred), *t4 = reinterpret_cast<const float4 *>(target);
    float4 *g4 = reinterpret_cast<float4 *>(grad);
    for (long long i = (long long)blockIdx.x * MSE_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * MSE_BLOCK) {
        const float4 a = p4[i], b = t4[i];
        g4[i] = make_float4(c * (a.x - b.x), c * (a.y - b.y), c * (a.z - b.z), c * (a.w - b.w));
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;
        if (i < n) grad[i] = c * (pred[i] - target[i]);
    }
}

static int mse_groups(long long n, int cap) { long long g = (n / 4 + MSE_BLOCK - 1) / MSE_BLOCK; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace gsr

extern "C" {

__attribute__((visibility("default"))) size_t gsr_mse_scratch_bytes(void) { return (gsr::MSE_MAX_GROUPS + (gsr::MSE_MAX_CLUSTERS + 1) * gsr::MSE_TICKET_STRIDE) * 4; }

__attribute__((visibility("default"))) int gsr_mse_forward(const float *pred, const float *target, int64_t n, float weight,
                                                           void *scratch, float *loss, void *stream)
{
    if (!pred || !target || !scratch || !loss || n <= 0) return GSR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) return GSR_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(gsr::k_mse_fwd, dim3(gsr::mse_groups(n, gsr::MSE_FWD_GROUPS)), dim3(gsr::MSE_BLOCK), 0, static_cast<hipStream_t>(stream), pred,
                       target, (long long)n, weight, static_cast<float *>(scratch), loss);
    return gsr::launch_status();
}

__attribute__((visibility("default"))) int gsr_mse_backward(const float *pred, const float *target, const float *grad_loss,
                                                            int64_t n, float weight, float *grad_pred, void *stream)
{
    if (!pred || !target || !grad_loss || !grad_pred || n <= 0) return GSR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad_pred)) & 15)
        return GSR_EINVAL;
    (void)hipGetLastError();
    hipLaunchKernelGGL(gsr::k_mse_bwd, dim3(gsr::mse_groups(n, gsr::MSE_MAX_GROUPS)), dim3(gsr::MSE_BLOCK), 0, static_cast<hipStream_t>(stream), pred,
                       target, grad_loss, (long long)n, weight, grad_pred);
    return gsr::launch_status();
}

}  // extern "C"
