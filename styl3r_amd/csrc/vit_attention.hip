// vit_attention.hip -- fp32 flash attention forward for gfx950 (head_dim 64, no mask).
//
// The reference runs attention in fp32 (xformers memory_efficient_attention on fp32 tensors,
// blocks.py:129,195); gfx950 has no TF32/xf32 MFMA, but it has the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 at the fp32 vector peak (157 TF).  Both contractions run on it, in the
// TRANSPOSED orientation so that the query index is always the MFMA column = the lane:
//
//     S^T (32 keys x 32 queries) = K (32 x 64) . Q^T (64 x 32)       A = K from LDS, B = Q in registers
//     O^T (32 d    x 32 queries) = V^T (32 x keys) . P^T (keys x 32)  A = V from LDS, B = P in registers
//
// With C/D layout col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5), the probabilities a lane holds
// after the softmax are exactly the B-operand fragments of the second product (no cross-lane
// movement), and the online-softmax rescale of O^T is lane-local (one query per lane; the two
// half-waves of a query exchange their partial max / sum once per tile).
// A workgroup = 4 wavefronts = 128 queries of one (batch, head); K/V tiles of 64 keys are staged in
// LDS once per workgroup (K with a 65-float row stride: the A-fragment read K[key=lane][d] is then
// bank-conflict free).  Optional fused 2-D RoPE: Q is rotated in registers, K while it is written to
// LDS, so the qkv buffer is never rewritten (the reference's curope pass costs 2 R/W sweeps of q and k).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;          // head dim
constexpr int QW = 32;          // queries per wavefront
constexpr int QB = 128;         // queries per workgroup
constexpr int KT = 64;          // keys per tile
constexpr int KSTR = 65;        // LDS row stride of the K tile (floats)

__device__ inline float wave_xor32(float x)
{
    // value of lane ^ 32 (the other half-wave of the same query)
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    // after the swap: x = [x.lo, y.lo] , y = [x.hi, y.hi] with y == old x  ->  lanes<32 read y (= x.hi), lanes>=32 read x (= x.lo)
    return (threadIdx.x & 32) ? x : y;
}

template <bool ROPE>
__global__ void __launch_bounds__(256) k_attn_fwd(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                  const float *__restrict__ v, float *__restrict__ out,
                                                  float *__restrict__ lse)
{
    __shared__ float s_k[KT * KSTR];
    __shared__ __attribute__((aligned(16))) float s_v[KT * HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * QB + wave * QW;
    const int qi = min(q0 + col, a.Nq - 1);   // clamped: rows beyond Nq compute garbage that is never stored
    const bool wave_active = q0 < a.Nq;       // N = 257 leaves most waves of the last workgroup without queries:
                                              // they only help staging the K/V tiles and skip the MFMA work
    const float qscale = a.scale * 1.4426950408889634f;   // scores in the base-2 domain

    // ---- Q fragment: qf[s] = Q[qi][2s + half], optionally rotated, pre-scaled ----
    float qf[32];
    {
        const float *qr = q + (int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh;
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] = qr[2 * s + half];
        if (ROPE) {
            const int64_t py = a.qpos[((int64_t)b * a.Nq + qi) * 2 + 0], px = a.qpos[((int64_t)b * a.Nq + qi) * 2 + 1];
            // feature d = 2s+half; quarters of 16: pairs (d, d+16) within [0,32) use py, within [32,64) use px
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int d = 2 * s + half;   // 0..15
                const float cy = a.cos_tab[py * 16 + d], sy = a.sin_tab[py * 16 + d];
                const float cx = a.cos_tab[px * 16 + d], sx = a.sin_tab[px * 16 + d];
                const float uy = qf[s], vy = qf[s + 8], ux = qf[s + 16], vx = qf[s + 24];
                qf[s] = uy * cy - vy * sy;      qf[s + 8] = vy * cy + uy * sy;
                qf[s + 16] = ux * cx - vx * sx; qf[s + 24] = vx * cx + ux * sx;
            }
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] *= qscale;
    }

    f32x16 o0 = {0}, o1 = {0};          // O^T rows d = rowmap(r) and 32 + rowmap(r), column = this lane's query
    float m = -INFINITY, l = 0.f;

    const float *kb = k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const float *vb = v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;

    // K/V tiles travel global -> registers -> LDS; the loads of tile t+1 are issued right after tile t has been
    // written to LDS, so their latency hides under tile t's MFMAs.
    float kreg[4][4];
    float4 vreg[4];
    int kpy[4], kpx[4];
    auto fetch = [&](int k0) {   // loads only: nothing here consumes a loaded value, so nothing waits for it
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + 256 * it;
            const int key = item >> 4, dq = item & 15;
            const int kg = min(k0 + key, a.Nk - 1);
            const float *kr = kb + (int64_t)kg * a.k_sn;
            kreg[it][0] = kr[dq]; kreg[it][1] = kr[16 + dq]; kreg[it][2] = kr[32 + dq]; kreg[it][3] = kr[48 + dq];
            vreg[it] = *reinterpret_cast<const float4 *>(vb + (int64_t)kg * a.v_sn + 4 * dq);
            if (ROPE) {
                kpy[it] = (int)a.kpos[((int64_t)b * a.Nk + kg) * 2 + 0];
                kpx[it] = (int)a.kpos[((int64_t)b * a.Nk + kg) * 2 + 1];
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < a.Nk; k0 += KT) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + 256 * it;
            const int key = item >> 4, dq = item & 15;
            const bool real = k0 + key < a.Nk;     // rows beyond Nk are written as zeros (and masked to -inf below)
            float uy = real ? kreg[it][0] : 0.f, vy = real ? kreg[it][1] : 0.f;
            float ux = real ? kreg[it][2] : 0.f, vx = real ? kreg[it][3] : 0.f;
            if (ROPE) {
                const float cy = a.cos_tab[kpy[it] * 16 + dq], sy = a.sin_tab[kpy[it] * 16 + dq];
                const float cx = a.cos_tab[kpx[it] * 16 + dq], sx = a.sin_tab[kpx[it] * 16 + dq];
                const float t0 = uy * cy - vy * sy, t1 = vy * cy + uy * sy;
                const float t2 = ux * cx - vx * sx, t3 = vx * cx + ux * sx;
                uy = t0; vy = t1; ux = t2; vx = t3;
            }
            float *dst = s_k + key * KSTR;
            dst[dq] = uy; dst[16 + dq] = vy; dst[32 + dq] = ux; dst[48 + dq] = vx;
            *reinterpret_cast<float4 *>(s_v + key * HD + 4 * dq) = real ? vreg[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (k0 + KT < a.Nk) fetch(k0 + KT);
        if (!wave_active) continue;
        const bool two = k0 + 32 < a.Nk;   // second 32-key block holds at least one real key

        // ---- S^T = K Q^T for the two 32-key blocks ----
        f32x16 st0 = {0}, st1 = {0};
        {
            const float *ka = s_k + col * KSTR + half;          // K[key = col][d = 2s + half]
            const float *kc = s_k + (32 + col) * KSTR + half;
            if (two) {
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    st0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], st0, 0, 0, 0);
                    st1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[2 * s], qf[s], st1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 32; ++s) st0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], st0, 0, 0, 0);
            }
        }
        // mask keys beyond Nk: element r of block kb is key k0 + 32 kb + (r&3) + 8 (r>>2) + 4 half
        if (k0 + KT > a.Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= a.Nk) st0[r] = -INFINITY;
                if (key + 32 >= a.Nk) st1[r] = -INFINITY;
            }
        }
        // ---- online softmax (base 2) ----
        float tmax = st0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st1[r]);
        tmax = fmaxf(tmax, wave_xor32(tmax));
        const float m_new = fmaxf(m, tmax);
        const float alpha = exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st0[r] = exp2f(st0[r] - m_new); psum += st0[r]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { st1[r] = exp2f(st1[r] - m_new); psum += st1[r]; }
        psum += wave_xor32(psum);
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T P^T : step (kb, r) contracts keys 32 kb + (r&3) + 8 (r>>2) + {0, 4} ----
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float *va = s_v + key * HD + col;           // V[key][d = col (+32)]
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], st0[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], st0[r], o1, 0, 0, 0);
            }
            if (two) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float *vc = s_v + key * HD + col;
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[0], st1[r], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[32], st1[r], o1, 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: O = O^T / l, out[q][d], d = 8g + 4 half + {0..3} (+32) ----
    if (q0 + col < a.Nq) {
        const float inv = 1.f / l;
        float *orow = out + (int64_t)b * a.o_sb + (int64_t)(q0 + col) * a.o_sn + (int64_t)h * a.o_sh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * half;
            *reinterpret_cast<float4 *>(orow + d) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + d) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
        if (lse && half == 0) lse[((int64_t)b * a.H + h) * a.Nq + q0 + col] = (m + log2f(l)) * 0.6931471805599453f;
    }
}

int attention_tail_rows(int n_rows, int n_other, int heads_times_batch);
hipError_t launch_attention_fwd_x6(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse, dim3 grid,
                                   int products, hipStream_t stream);

// 0: both contractions on the exact-f32 MFMA (this file); 1 (default): bf16x6 split arithmetic on the bf16 MFMA (vit_attention_x6.hip);
// 2: the same kernels with three partial products per contraction step ("bf16x3": operands good to 2^-18, as vit_x6_set_products(3)).
// Per HOST THREAD (thread_local, like vit_x6_set_products): a thread's set + launch pair cannot be interleaved with another
// thread's choice; read at launch time on the launching thread.
static thread_local int g_attn_arith = 1;
int attention_set_arith(int mode)
{
    if (mode < 0 || mode > 3) return VIT_EINVAL;
    g_attn_arith = mode;
    return VIT_OK;
}
int attention_arith() { return g_attn_arith; }
int attention_fwd_tail(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse, int rows,
                       hipStream_t stream);
int amax(const float *x, int64_t n, void *out, hipStream_t stream);      // vit_gemm_x6.hip

int attention_fwd(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse,
                  hipStream_t stream)
{
    if (!q || !k || !v || !out) return VIT_EINVAL;
    if (a.B <= 0 || a.H <= 0 || a.Nq <= 0 || a.Nk <= 0) return VIT_EINVAL;
    const bool rope = a.cos_tab != nullptr;
    if (rope && (!a.sin_tab || !a.qpos || !a.kpos || a.P <= 0)) return VIT_EINVAL;
    // float4 epilogue / V loads need 16-byte aligned rows
    if ((a.o_sn | a.o_sh | a.o_sb | a.v_sn | a.v_sh | a.v_sb) & 3) return VIT_EINVAL;
    // 257 = 2 x 128 + 1 queries: the last 1..4 rows go to the vector kernel (vit_attention_tail.hip) instead of a third,
    // almost empty workgroup per (batch, head) that would open a second round on the chip
    const int tail_rows = attention_tail_rows(a.Nq, a.Nk, a.H * a.B);
    const dim3 grid((a.Nq - tail_rows + QB - 1) / QB, a.H, a.B);
    (void)hipGetLastError();
    // the split-arithmetic kernel loads q / k rows as float4: strides in multiples of 4 floats, 16-byte aligned bases
    const bool x6_ok = !((a.q_sn | a.q_sh | a.q_sb | a.k_sn | a.k_sh | a.k_sb) & 3) && !((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15);
    hipError_t e;
    if (attention_arith() == 3 && x6_ok && !(a.amax_q && a.amax_k && a.amax_v)) return VIT_EINVAL;     // f16x3 without the operands' |max| words: refuse, never guess a scale
    if (attention_arith() >= 1 && x6_ok) e = launch_attention_fwd_x6(a, q, k, v, out, lse, grid, attention_arith() == 2 ? 3 : (attention_arith() == 3 ? 2 : 6), stream);
    else {
        // the exact-f32 kernel has no |max| epilogue: a requested word is filled by a pass over the (contiguous) result instead
        VitAttnArgs b = a;
        b.amax_out = nullptr;
        const bool contiguous = a.o_sh == HD && a.o_sn == (int64_t)a.H * HD && a.o_sb == (int64_t)a.Nq * a.H * HD;
        if (a.amax_out && !contiguous) return VIT_EINVAL;
        if (rope) hipLaunchKernelGGL(k_attn_fwd<true>, grid, dim3(256), 0, stream, b, q, k, v, out, lse);
        else hipLaunchKernelGGL(k_attn_fwd<false>, grid, dim3(256), 0, stream, b, q, k, v, out, lse);
        e = hipGetLastError();
        if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
        if (tail_rows) { const int rc = attention_fwd_tail(b, q, k, v, out, lse, tail_rows, stream); if (rc != VIT_OK) return rc; }
        return a.amax_out ? amax(out, (int64_t)a.B * a.Nq * a.H * HD, a.amax_out, stream) : VIT_OK;
    }
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    if (tail_rows) return attention_fwd_tail(a, q, k, v, out, lse, tail_rows, stream);
    return VIT_OK;
}
}  // namespace vit
