// vit_attention_tail.hip -- the few leftover rows of an attention call whose length is a multiple of 128 plus 1..4.
//
// The CroCo encoder appends ONE intrinsics token to the 256 (or 1024) patch tokens of a view, so almost every attention
// in the model has 257 queries and keys.  The tiled kernels give 128 rows to a workgroup: the 257th row would cost a
// third workgroup per (batch, head) -- 768 instead of 512 workgroups on a chip that holds 512, i.e. a second, nearly
// empty round: measured +46 % (forward) / +63 % (backward) for one extra token.  These kernels compute that row (a
// vector-matrix problem, no MFMA) with one wavefront per (row, head, batch); the tiled kernels then launch only their
// full 128-row blocks.  Same conventions as the tiled kernels: 2-D RoPE fused into the loads (feature pairs (d, d+16)
// of each 32-feature half, first half by y, second by x), scores in the base-2 domain, lse in natural units.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vit_ops.h"
#include "vit_amax.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

namespace tail {
constexpr int HD = 64, MAXN = 8192;      // rows of the OTHER operand that fit the score buffer

// rotate one feature of a 64-vector held in LDS (raw[]) by the 2-D position (py, px): returns the rotated feature d
__device__ inline float rope_feature(const float *raw, int d, int64_t py, int64_t px, const float *cos_tab, const float *sin_tab,
                                     float sign)
{
    const int g = d >> 4, dq = d & 15;                 // g: 0,1 -> y pair (u, v) ; 2,3 -> x pair
    const int64_t pos = g < 2 ? py : px;
    const float c = cos_tab[pos * 16 + dq], s = sign * sin_tab[pos * 16 + dq];
    const float u = raw[(g & 2) * 16 + dq], v = raw[(g & 2) * 16 + 16 + dq];
    return (g & 1) ? (v * c + u * s) : (u * c - v * s);
}

// rotate a whole 64-vector held in registers (x[d]) in place
__device__ inline void rope_regs(float *x, int64_t py, int64_t px, const float *cos_tab, const float *sin_tab)
{
#pragma unroll
    for (int dq = 0; dq < 16; ++dq) {
        const float cy = cos_tab[py * 16 + dq], sy = sin_tab[py * 16 + dq];
        const float cx = cos_tab[px * 16 + dq], sx = sin_tab[px * 16 + dq];
        const float uy = x[dq], vy = x[16 + dq], ux = x[32 + dq], vx = x[48 + dq];
        x[dq] = uy * cy - vy * sy; x[16 + dq] = vy * cy + uy * sy;
        x[32 + dq] = ux * cx - vx * sx; x[48 + dq] = vx * cx + ux * sx;
    }
}

__device__ inline float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- forward: out[qi] = softmax(q_qi K^T * scale) V, lse[qi] ----
template <bool ROPE>
__global__ void __launch_bounds__(64) k_attn_fwd_tail(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ v, float *__restrict__ out,
                                                      float *__restrict__ lse, int q_first)
{
    __shared__ float s_raw[HD], s_q[HD], s_sc[MAXN];
    const int lane = threadIdx.x, h = blockIdx.y, b = blockIdx.z, qi = q_first + blockIdx.x;
    const float qscale = a.scale * 1.4426950408889634f;
    s_raw[lane] = q[(int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh + lane];
    __syncthreads();
    float qv = s_raw[lane];
    if (ROPE) qv = rope_feature(s_raw, lane, a.qpos[((int64_t)b * a.Nq + qi) * 2], a.qpos[((int64_t)b * a.Nq + qi) * 2 + 1], a.cos_tab, a.sin_tab, 1.f);
    s_q[lane] = qv * qscale;
    __syncthreads();
    // pass 1: lane = key, scores to LDS
    float mx = -INFINITY;
    for (int j = lane; j < a.Nk; j += 64) {
        float kk[HD];
        const float4 *kr = reinterpret_cast<const float4 *>(k + (int64_t)b * a.k_sb + (int64_t)j * a.k_sn + (int64_t)h * a.k_sh);
#pragma unroll
        for (int t = 0; t < 16; ++t) { const float4 x = kr[t]; kk[4 * t] = x.x; kk[4 * t + 1] = x.y; kk[4 * t + 2] = x.z; kk[4 * t + 3] = x.w; }
        if (ROPE) rope_regs(kk, a.kpos[((int64_t)b * a.Nk + j) * 2], a.kpos[((int64_t)b * a.Nk + j) * 2 + 1], a.cos_tab, a.sin_tab);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += kk[d] * s_q[d];
        s_sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    __syncthreads();
    // pass 2: lane = output feature
    float l = 0.f, o = 0.f;
    const float *vb = v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh + lane;
    int j = 0;
    for (; j + 16 <= a.Nk; j += 16) {                      // sixteen independent row loads in flight
        float vv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) vv[u] = vb[(int64_t)(j + u) * a.v_sn];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float p = exp2f(s_sc[j + u] - mx);
            l += p;
            o += p * vv[u];
        }
    }
    for (; j < a.Nk; ++j) {
        const float p = exp2f(s_sc[j] - mx);
        l += p;
        o += p * vb[(int64_t)j * a.v_sn];
    }
    out[(int64_t)b * a.o_sb + (int64_t)qi * a.o_sn + (int64_t)h * a.o_sh + lane] = o / l;
    if (a.amax_out) amax_word_fold(a.amax_out, abs_bits(o / l));
    if (lse && lane == 0) lse[((int64_t)b * a.H + h) * a.Nq + qi] = (mx + log2f(l)) * 0.6931471805599453f;
}

// ---- backward, query side: dQ[qi] = sum_j dS_j k_j ; dS_j = P_j (dO_qi . v_j - delta_qi) scale ----
// dout, dq contiguous (B, Nq, H, 64) as in the tiled kernels
template <bool ROPE>
__global__ void __launch_bounds__(64) k_attn_bwd_q_tail(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, const float *__restrict__ dout,
                                                        const float *__restrict__ lse, const float *__restrict__ delta,
                                                        float *__restrict__ dq, int q_first)
{
    __shared__ float s_raw[HD], s_q[HD], s_do[HD], s_acc[64 * 65];
    const int lane = threadIdx.x, h = blockIdx.y, b = blockIdx.z, qi = q_first + blockIdx.x;
    const int64_t py = ROPE ? a.qpos[((int64_t)b * a.Nq + qi) * 2] : 0, px = ROPE ? a.qpos[((int64_t)b * a.Nq + qi) * 2 + 1] : 0;
    s_raw[lane] = q[(int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh + lane];
    s_do[lane] = dout[(((int64_t)b * a.Nq + qi) * a.H + h) * HD + lane];
    __syncthreads();
    float qv = s_raw[lane];
    if (ROPE) qv = rope_feature(s_raw, lane, py, px, a.cos_tab, a.sin_tab, 1.f);
    s_q[lane] = qv * (a.scale * 1.4426950408889634f);
    __syncthreads();
    const float lse2 = lse[((int64_t)b * a.H + h) * a.Nq + qi] * 1.4426950408889634f;
    const float del = delta[((int64_t)b * a.H + h) * a.Nq + qi];
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = lane; j < a.Nk; j += 64) {
        float kk[HD];
        const float4 *kr = reinterpret_cast<const float4 *>(k + (int64_t)b * a.k_sb + (int64_t)j * a.k_sn + (int64_t)h * a.k_sh);
        const float4 *vr = reinterpret_cast<const float4 *>(v + (int64_t)b * a.v_sb + (int64_t)j * a.v_sn + (int64_t)h * a.v_sh);
        float dp = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 x = kr[t]; kk[4 * t] = x.x; kk[4 * t + 1] = x.y; kk[4 * t + 2] = x.z; kk[4 * t + 3] = x.w;
            const float4 y = vr[t];
            dp += y.x * s_do[4 * t] + y.y * s_do[4 * t + 1] + y.z * s_do[4 * t + 2] + y.w * s_do[4 * t + 3];
        }
        if (ROPE) rope_regs(kk, a.kpos[((int64_t)b * a.Nk + j) * 2], a.kpos[((int64_t)b * a.Nk + j) * 2 + 1], a.cos_tab, a.sin_tab);
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) sc += kk[d] * s_q[d];
        const float ds = exp2f(sc - lse2) * (dp - del) * a.scale;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += ds * kk[d];
    }
    // sum the per-lane partials over the wavefront: lane d adds column d of the 64 x 64 table
#pragma unroll
    for (int d = 0; d < HD; ++d) s_acc[lane * 65 + d] = acc[d];
    __syncthreads();
    float tot = 0.f;
#pragma unroll 8
    for (int l2 = 0; l2 < 64; ++l2) tot += s_acc[l2 * 65 + lane];
    s_raw[lane] = tot;                          // dQ w.r.t. the rotated query
    __syncthreads();
    float r = tot;
    if (ROPE) r = rope_feature(s_raw, lane, py, px, a.cos_tab, a.sin_tab, -1.f);   // inverse rotation
    dq[((int64_t)b * a.Nq + qi) * (a.dq_sn ? a.dq_sn : (int64_t)a.H * HD) + h * HD + lane] = r;
    if (a.amax_dq) amax_word_fold(a.amax_dq, abs_bits(r));
}

// ---- backward, key side: dK[kj] = sum_i dS_i q_i , dV[kj] = sum_i P_i dO_i over ALL queries i ----
template <bool ROPE>
__global__ void __launch_bounds__(64) k_attn_bwd_kv_tail(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                         const float *__restrict__ v, const float *__restrict__ dout,
                                                         const float *__restrict__ lse, const float *__restrict__ delta,
                                                         float *__restrict__ dk, float *__restrict__ dv, int k_first)
{
    __shared__ float s_raw[HD], s_k[HD], s_v[HD], s_acc[64 * 65];
    const int lane = threadIdx.x, h = blockIdx.y, b = blockIdx.z, kj = k_first + blockIdx.x;
    const int64_t py = ROPE ? a.kpos[((int64_t)b * a.Nk + kj) * 2] : 0, px = ROPE ? a.kpos[((int64_t)b * a.Nk + kj) * 2 + 1] : 0;
    s_raw[lane] = k[(int64_t)b * a.k_sb + (int64_t)kj * a.k_sn + (int64_t)h * a.k_sh + lane];
    s_v[lane] = v[(int64_t)b * a.v_sb + (int64_t)kj * a.v_sn + (int64_t)h * a.v_sh + lane];
    __syncthreads();
    float kv = s_raw[lane];
    if (ROPE) kv = rope_feature(s_raw, lane, py, px, a.cos_tab, a.sin_tab, 1.f);
    s_k[lane] = kv * (a.scale * 1.4426950408889634f);
    __syncthreads();
    float acck[HD], accv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { acck[d] = 0.f; accv[d] = 0.f; }
    for (int i = lane; i < a.Nq; i += 64) {
        float qq[HD];
        const float4 *qr = reinterpret_cast<const float4 *>(q + (int64_t)b * a.q_sb + (int64_t)i * a.q_sn + (int64_t)h * a.q_sh);
        const float4 *gr = reinterpret_cast<const float4 *>(dout + (((int64_t)b * a.Nq + i) * a.H + h) * HD);
        float dp = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 x = qr[t]; qq[4 * t] = x.x; qq[4 * t + 1] = x.y; qq[4 * t + 2] = x.z; qq[4 * t + 3] = x.w;
            const float4 y = gr[t];
            dp += y.x * s_v[4 * t] + y.y * s_v[4 * t + 1] + y.z * s_v[4 * t + 2] + y.w * s_v[4 * t + 3];
        }
        if (ROPE) rope_regs(qq, a.qpos[((int64_t)b * a.Nq + i) * 2], a.qpos[((int64_t)b * a.Nq + i) * 2 + 1], a.cos_tab, a.sin_tab);
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) sc += qq[d] * s_k[d];
        const float p = exp2f(sc - lse[((int64_t)b * a.H + h) * a.Nq + i] * 1.4426950408889634f);
        const float ds = p * (dp - delta[((int64_t)b * a.H + h) * a.Nq + i]) * a.scale;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 y = gr[t];                        // (second read of dO_i: L1/L2 resident)
            accv[4 * t] += p * y.x; accv[4 * t + 1] += p * y.y; accv[4 * t + 2] += p * y.z; accv[4 * t + 3] += p * y.w;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) acck[d] += ds * qq[d];
    }
    // dV
#pragma unroll
    for (int d = 0; d < HD; ++d) s_acc[lane * 65 + d] = accv[d];
    __syncthreads();
    float tv = 0.f;
#pragma unroll 8
    for (int l2 = 0; l2 < 64; ++l2) tv += s_acc[l2 * 65 + lane];
    dv[((int64_t)b * a.Nk + kj) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD + lane] = tv;
    if (a.amax_dv) amax_word_fold(a.amax_dv, abs_bits(tv));
    __syncthreads();
    // dK (w.r.t. the rotated key), then rotate back
#pragma unroll
    for (int d = 0; d < HD; ++d) s_acc[lane * 65 + d] = acck[d];
    __syncthreads();
    float tk = 0.f;
#pragma unroll 8
    for (int l2 = 0; l2 < 64; ++l2) tk += s_acc[l2 * 65 + lane];
    s_raw[lane] = tk;
    __syncthreads();
    float r = tk;
    if (ROPE) r = rope_feature(s_raw, lane, py, px, a.cos_tab, a.sin_tab, -1.f);
    dk[((int64_t)b * a.Nk + kj) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD + lane] = r;
    if (a.amax_dk) amax_word_fold(a.amax_dk, abs_bits(r));
}
}  // namespace tail

// how many trailing rows the tail kernels take over (0: none, the tiled kernel covers everything).  They pay off only when
// dropping the ragged last block saves a whole ROUND of workgroups on the chip (512 resident: 256 CUs x 2); on small grids
// (batch-1 inference, the per-view decoders) the extra block is free and the tail launches would only add latency.
int attention_tail_rows(int n_rows, int n_other, int heads_times_batch)
{
    const int r = n_rows % 128;
    if (!(n_rows > 128 && r >= 1 && r <= 4 && n_other <= tail::MAXN)) return 0;
    const long long with_block = (long long)((n_rows + 127) / 128) * heads_times_batch, without = (long long)(n_rows / 128) * heads_times_batch;
    return (with_block + 511) / 512 > (without + 511) / 512 ? r : 0;
}

int attention_fwd_tail(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse, int rows,
                       hipStream_t stream)
{
    const dim3 grid(rows, a.H, a.B);
    if (a.cos_tab) hipLaunchKernelGGL(tail::k_attn_fwd_tail<true>, grid, dim3(64), 0, stream, a, q, k, v, out, lse, a.Nq - rows);
    else hipLaunchKernelGGL(tail::k_attn_fwd_tail<false>, grid, dim3(64), 0, stream, a, q, k, v, out, lse, a.Nq - rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
int attention_bwd_tails(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *lse, const float *dout,
                        const float *delta, float *dq, float *dk, float *dv, int q_rows, int k_rows, hipStream_t stream)
{
    const bool rope = a.cos_tab != nullptr;
    if (q_rows) {
        const dim3 g(q_rows, a.H, a.B);
        if (rope) hipLaunchKernelGGL(tail::k_attn_bwd_q_tail<true>, g, dim3(64), 0, stream, a, q, k, v, dout, lse, delta, dq, a.Nq - q_rows);
        else hipLaunchKernelGGL(tail::k_attn_bwd_q_tail<false>, g, dim3(64), 0, stream, a, q, k, v, dout, lse, delta, dq, a.Nq - q_rows);
    }
    if (k_rows) {
        const dim3 g(k_rows, a.H, a.B);
        if (rope) hipLaunchKernelGGL(tail::k_attn_bwd_kv_tail<true>, g, dim3(64), 0, stream, a, q, k, v, dout, lse, delta, dk, dv, a.Nk - k_rows);
        else hipLaunchKernelGGL(tail::k_attn_bwd_kv_tail<false>, g, dim3(64), 0, stream, a, q, k, v, dout, lse, delta, dk, dv, a.Nk - k_rows);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
