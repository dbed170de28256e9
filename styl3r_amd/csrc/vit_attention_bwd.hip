// vit_attention_bwd.hip -- fp32 flash attention backward for gfx950 (head_dim 64, no mask).
//
// Two passes, no atomics, every contraction on v_mfma_f32_32x32x2_f32 (exact f32):
//   k_attn_delta   delta[b,h,q] = sum_d dO[q,d] O[q,d]
//   k_attn_bwd_kv  a wavefront owns 32 KEYS (key = MFMA column = lane) and walks the query tiles:
//                    S  = Q K^T         A = Q  (LDS)   B = K  (regs)   -> P = exp2(S2 - lse2[q])
//                    dP = dO V^T        A = dO (LDS)   B = V  (regs)
//                    dV^T += dO^T P     A = dO^T (LDS) B = P  (regs, as produced)
//                    dK^T += Q^T dS     A = Q^T  (LDS) B = dS (regs, as produced)
//   k_attn_bwd_q   a wavefront owns 32 QUERIES (query = lane) and walks the key tiles:
//                    S^T = K Q^T, dP^T = V dO^T (A from LDS, B = Q / dO in regs), dQ^T += K^T dS^T
// In both passes the probabilities come out of the first MFMA already in the B-operand layout of the
// MFMA that consumes them, so nothing moves between lanes.  One LDS tile with a 65-float row stride
// serves both the "row = lane" and the "column = lane" A-fragment reads without bank conflicts.
// With fused RoPE, Q and K are rotated while staged / loaded and dQ, dK are rotated back (lane-local:
// features d and d+16 sit in registers r and r+8 of the same lane) before they are stored.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HD = 64, TSTR = 65, TROWS = 32;
constexpr float LOG2E = 1.4426950408889634f;

__device__ inline int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ------------------------------------------------------------------ delta
__global__ void __launch_bounds__(256) k_attn_delta(const float *__restrict__ o, const float *__restrict__ g, float *__restrict__ delta,
                                                    int B, int H, int Nq)
{
    // o, g contiguous (B, Nq, H, 64); one 16-lane group per (b,q,h) row
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const long long rows = (long long)B * Nq * H;
    float s = 0.f;
    if (row < rows) {
        const float4 a = reinterpret_cast<const float4 *>(o + row * HD)[sub];
        const float4 c = reinterpret_cast<const float4 *>(g + row * HD)[sub];
        s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
    if (row < rows && sub == 0) {
        const int h = (int)(row % H);
        const long long bq = row / H;
        const int qi = (int)(bq % Nq);
        const long long b = bq / Nq;
        delta[(b * H + h) * Nq + qi] = s;
    }
}

// stage TROWS rows x 64 floats (rotated if ROPE) from a strided (b, n, h) tensor into LDS, stride TSTR; rows >= n_valid are zero
template <bool ROPE>
__device__ inline void stage_rows(float *__restrict__ dst, const float *__restrict__ base, int64_t sn, int row0, int n_valid,
                                  const int64_t *__restrict__ pos, const float *__restrict__ cos_tab,
                                  const float *__restrict__ sin_tab, int tid)
{
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + 256 * it;
        const int row = item >> 4, dq = item & 15;
        const int gi = row0 + row;
        float uy = 0.f, vy = 0.f, ux = 0.f, vx = 0.f;
        if (gi < n_valid) {
            const float *rp = base + (int64_t)gi * sn;
            uy = rp[dq]; vy = rp[16 + dq]; ux = rp[32 + dq]; vx = rp[48 + dq];
            if (ROPE) {
                const int64_t py = pos[(int64_t)gi * 2 + 0], px = pos[(int64_t)gi * 2 + 1];
                const float cy = cos_tab[py * 16 + dq], sy = sin_tab[py * 16 + dq];
                const float cx = cos_tab[px * 16 + dq], sx = sin_tab[px * 16 + dq];
                const float t0 = uy * cy - vy * sy, t1 = vy * cy + uy * sy;
                const float t2 = ux * cx - vx * sx, t3 = vx * cx + ux * sx;
                uy = t0; vy = t1; ux = t2; vx = t3;
            }
        }
        float *d = dst + row * TSTR;
        d[dq] = uy; d[16 + dq] = vy; d[32 + dq] = ux; d[48 + dq] = vx;
    }
}

// load this lane's B-fragment of a row: f[s] = X[row][2s + half], rotated if ROPE
template <bool ROPE>
__device__ inline void load_frag(float *f, const float *__restrict__ rp, int half, const int64_t *__restrict__ pos2,
                                 const float *__restrict__ cos_tab, const float *__restrict__ sin_tab)
{
#pragma unroll
    for (int s = 0; s < 32; ++s) f[s] = rp[2 * s + half];
    if (ROPE) {
        const int64_t py = pos2[0], px = pos2[1];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int d = 2 * s + half;
            const float cy = cos_tab[py * 16 + d], sy = sin_tab[py * 16 + d];
            const float cx = cos_tab[px * 16 + d], sx = sin_tab[px * 16 + d];
            const float uy = f[s], vy = f[s + 8], ux = f[s + 16], vx = f[s + 24];
            f[s] = uy * cy - vy * sy;      f[s + 8] = vy * cy + uy * sy;
            f[s + 16] = ux * cx - vx * sx; f[s + 24] = vx * cx + ux * sx;
        }
    }
}

// inverse rotation of a transposed 64 x (lane) gradient held as two f32x16 (rows rowmap(r) and 32 + rowmap(r)):
// features d < 16 pair with d + 16 -> registers r < 8 pair with r + 8 of the same accumulator.
__device__ inline void unrotate(f32x16 &lo, f32x16 &hi, int half, int64_t py, int64_t px, const float *__restrict__ cos_tab,
                                const float *__restrict__ sin_tab)
{
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int d = rowmap(r, half);   // 0..15
        const float cy = cos_tab[py * 16 + d], sy = sin_tab[py * 16 + d];
        const float cx = cos_tab[px * 16 + d], sx = sin_tab[px * 16 + d];
        const float gu = lo[r], gv = lo[r + 8];
        lo[r] = gu * cy + gv * sy; lo[r + 8] = gv * cy - gu * sy;      // transpose of [[c,-s],[s,c]]
        const float hu = hi[r], hv = hi[r + 8];
        hi[r] = hu * cx + hv * sx; hi[r + 8] = hv * cx - hu * sx;
    }
}

// ------------------------------------------------------------------ dK, dV
template <bool ROPE>
__global__ void __launch_bounds__(256) k_attn_bwd_kv(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                     const float *__restrict__ v, const float *__restrict__ g,
                                                     const float *__restrict__ lse, const float *__restrict__ delta,
                                                     float *__restrict__ dk, float *__restrict__ dv)
{
    __shared__ float s_q[TROWS * TSTR], s_g[TROWS * TSTR];
    __shared__ float s_lse[TROWS], s_delta[TROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int key0 = blockIdx.x * 128 + wave * 32;
    const int ki = min(key0 + col, a.Nk - 1);
    const float c2 = a.scale * LOG2E;

    float kfs[32], vf[32];      // K fragment pre-scaled by scale*log2(e): only S = Q K^T consumes it
    load_frag<ROPE>(kfs, k + (int64_t)b * a.k_sb + (int64_t)ki * a.k_sn + (int64_t)h * a.k_sh, half,
                    ROPE ? a.kpos + ((int64_t)b * a.Nk + ki) * 2 : nullptr, a.cos_tab, a.sin_tab);
    load_frag<false>(vf, v + (int64_t)b * a.v_sb + (int64_t)ki * a.v_sn + (int64_t)h * a.v_sh, half, nullptr, nullptr, nullptr);
#pragma unroll
    for (int s = 0; s < 32; ++s) kfs[s] *= c2;

    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const float *qb = q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const float *gb = g + ((int64_t)b * a.Nq * a.H + h) * HD;          // contiguous (B,Nq,H,64)
    const int64_t g_sn = (int64_t)a.H * HD;
    const float *lse_b = lse + ((int64_t)b * a.H + h) * a.Nq;
    const float *del_b = delta + ((int64_t)b * a.H + h) * a.Nq;

    for (int q0 = 0; q0 < a.Nq; q0 += TROWS) {
        __syncthreads();
        stage_rows<ROPE>(s_q, qb, a.q_sn, q0, a.Nq, ROPE ? a.qpos + (int64_t)b * a.Nq * 2 : nullptr, a.cos_tab, a.sin_tab, tid);
        stage_rows<false>(s_g, gb, g_sn, q0, a.Nq, nullptr, nullptr, nullptr, tid);
        if (tid < TROWS) {
            const int qi = q0 + tid;
            s_lse[tid] = qi < a.Nq ? lse_b[qi] * LOG2E : INFINITY;   // padded queries: P = exp2(-inf) = 0
            s_delta[tid] = qi < a.Nq ? del_b[qi] : 0.f;
        }
        __syncthreads();

        f32x16 sc = {0}, dp = {0};
        {
            const float *qa = s_q + col * TSTR + half, *ga = s_g + col * TSTR + half;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * s], kfs[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * s], vf[s], dp, 0, 0, 0);
            }
        }
        // element r: query q0 + rowmap(r), key = this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = rowmap(r, half);
            const float p = exp2f(sc[r] - s_lse[qr]);
            sc[r] = p;                                        // P
            dp[r] = p * (dp[r] - s_delta[qr]) * a.scale;      // dS (w.r.t. the unscaled dot product)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = rowmap(r, half);
            const float *gr = s_g + qr * TSTR + col, *qr_ = s_q + qr * TSTR + col;
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[0], sc[r], dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[32], sc[r], dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qr_[0], dp[r], dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qr_[32], dp[r], dk1, 0, 0, 0);
        }
    }
    if (key0 + col < a.Nk) {
        if (ROPE) {
            const int64_t *pp = a.kpos + ((int64_t)b * a.Nk + key0 + col) * 2;
            unrotate(dk0, dk1, half, pp[0], pp[1], a.cos_tab, a.sin_tab);
        }
        // dk, dv contiguous (B,Nk,H,64)
        float *dkr = dk + ((int64_t)b * a.Nk + key0 + col) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD;
        float *dvr = dv + ((int64_t)b * a.Nk + key0 + col) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = 8 * gq + 4 * half;
            *reinterpret_cast<float4 *>(dkr + d) = make_float4(dk0[4 * gq], dk0[4 * gq + 1], dk0[4 * gq + 2], dk0[4 * gq + 3]);
            *reinterpret_cast<float4 *>(dkr + 32 + d) = make_float4(dk1[4 * gq], dk1[4 * gq + 1], dk1[4 * gq + 2], dk1[4 * gq + 3]);
            *reinterpret_cast<float4 *>(dvr + d) = make_float4(dv0[4 * gq], dv0[4 * gq + 1], dv0[4 * gq + 2], dv0[4 * gq + 3]);
            *reinterpret_cast<float4 *>(dvr + 32 + d) = make_float4(dv1[4 * gq], dv1[4 * gq + 1], dv1[4 * gq + 2], dv1[4 * gq + 3]);
        }
    }
}

// ------------------------------------------------------------------ dQ
template <bool ROPE>
__global__ void __launch_bounds__(256, 3) k_attn_bwd_q(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                    const float *__restrict__ v, const float *__restrict__ g,
                                                    const float *__restrict__ lse, const float *__restrict__ delta,
                                                    float *__restrict__ dq)
{
    __shared__ float s_k[TROWS * TSTR], s_v[TROWS * TSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = min(q0 + col, a.Nq - 1);
    const float c2 = a.scale * LOG2E;

    float qf[32], gf[32];
    load_frag<ROPE>(qf, q + (int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh, half,
                    ROPE ? a.qpos + ((int64_t)b * a.Nq + qi) * 2 : nullptr, a.cos_tab, a.sin_tab);
    load_frag<false>(gf, g + (((int64_t)b * a.Nq + qi) * a.H + h) * HD, half, nullptr, nullptr, nullptr);
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[s] *= c2;
    const float lse2 = lse[((int64_t)b * a.H + h) * a.Nq + qi] * LOG2E;
    const float del = delta[((int64_t)b * a.H + h) * a.Nq + qi];

    f32x16 dq0 = {0}, dq1 = {0};
    const float *kb = k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const float *vb = v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;

    for (int k0 = 0; k0 < a.Nk; k0 += TROWS) {
        __syncthreads();
        stage_rows<ROPE>(s_k, kb, a.k_sn, k0, a.Nk, ROPE ? a.kpos + (int64_t)b * a.Nk * 2 : nullptr, a.cos_tab, a.sin_tab, tid);
        stage_rows<false>(s_v, vb, a.v_sn, k0, a.Nk, nullptr, nullptr, nullptr, tid);
        __syncthreads();
        f32x16 st = {0}, dp = {0};
        {
            const float *ka = s_k + col * TSTR + half, *va = s_v + col * TSTR + half;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * s], gf[s], dp, 0, 0, 0);
            }
        }
        // element r: key k0 + rowmap(r), query = this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + rowmap(r, half);
            const float p = key < a.Nk ? exp2f(st[r] - lse2) : 0.f;
            dp[r] = p * (dp[r] - del) * a.scale;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *kr = s_k + rowmap(r, half) * TSTR + col;
            dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[0], dp[r], dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32], dp[r], dq1, 0, 0, 0);
        }
    }
    if (q0 + col < a.Nq) {
        if (ROPE) {
            const int64_t *pp = a.qpos + ((int64_t)b * a.Nq + q0 + col) * 2;
            unrotate(dq0, dq1, half, pp[0], pp[1], a.cos_tab, a.sin_tab);
        }
        float *dqr = dq + ((int64_t)b * a.Nq + q0 + col) * (a.dq_sn ? a.dq_sn : (int64_t)a.H * HD) + h * HD;   // (B,Nq,H,64), token stride dq_sn
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int d = 8 * gq + 4 * half;
            *reinterpret_cast<float4 *>(dqr + d) = make_float4(dq0[4 * gq], dq0[4 * gq + 1], dq0[4 * gq + 2], dq0[4 * gq + 3]);
            *reinterpret_cast<float4 *>(dqr + 32 + d) = make_float4(dq1[4 * gq], dq1[4 * gq + 1], dq1[4 * gq + 2], dq1[4 * gq + 3]);
        }
    }
}

int attention_tail_rows(int n_rows, int n_other, int heads_times_batch);
int attention_arith();
hipError_t launch_attention_bwd_x6(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *dout, const float *lse,
                                   const float *delta, float *dq, float *dk, float *dv, dim3 gkv, dim3 gq, int products, hipStream_t stream);
int attention_bwd_tails(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *lse, const float *dout,
                        const float *delta, float *dq, float *dk, float *dv, int q_rows, int k_rows, hipStream_t stream);

int amax(const float *x, int64_t n, void *out, hipStream_t stream);      // vit_gemm_x6.hip

int attention_bwd(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *out, const float *lse,
                  const float *dout, float *dq, float *dk, float *dv, float *delta_ws, hipStream_t stream)
{
    if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv || !delta_ws) return VIT_EINVAL;
    if (a.B <= 0 || a.H <= 0 || a.Nq <= 0 || a.Nk <= 0) return VIT_EINVAL;
    const bool rope = a.cos_tab != nullptr;
    if (rope && (!a.sin_tab || !a.qpos || !a.kpos || a.P <= 0)) return VIT_EINVAL;
    (void)hipGetLastError();
    const long long rows = (long long)a.B * a.Nq * a.H;
    hipLaunchKernelGGL(k_attn_delta, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, stream, out, dout, delta_ws, a.B, a.H, a.Nq);
    // the last 1..4 rows of a 128 n + r problem (257 tokens) go to the vector kernels of vit_attention_tail.hip: a third,
    // almost empty workgroup per (batch, head) would open a second round on the chip (+63 % measured)
    const int q_tail = attention_tail_rows(a.Nq, a.Nk, a.H * a.B), k_tail = attention_tail_rows(a.Nk, a.Nq, a.H * a.B);
    const dim3 gkv((a.Nk - k_tail + 127) / 128, a.H, a.B), gq((a.Nq - q_tail + 127) / 128, a.H, a.B);
    // the split-arithmetic kernels (vit_attention_bwd_x6.hip) load rows as float4: strides in multiples of 4 floats, 16-byte aligned bases
    const bool x6_ok = !((a.q_sn | a.q_sh | a.q_sb | a.k_sn | a.k_sh | a.k_sb | a.v_sn | a.v_sh | a.v_sb) & 3) &&
                       !((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout)) & 15);
    hipError_t e;
    if (attention_arith() == 3 && x6_ok && !(a.amax_q && a.amax_k && a.amax_v && a.amax_g)) return VIT_EINVAL;     // f16x3: the operands' |max| words are required
    if (attention_arith() >= 1 && x6_ok) {
        e = launch_attention_bwd_x6(a, q, k, v, dout, lse, delta_ws, dq, dk, dv, gkv, gq, attention_arith() == 2 ? 3 : (attention_arith() == 3 ? 2 : 6), stream);
    } else {
        // the exact-f32 kernels have no |max| epilogue: requested words are filled by passes over the results -- ONE pass over the packed
        // (B,N,3,H,64) gradient when the three words are one and dq / dk / dv are its planes, else one per contiguous tensor
        VitAttnArgs b = a;
        b.amax_dq = b.amax_dk = b.amax_dv = nullptr;
        const bool want = a.amax_dq || a.amax_dk || a.amax_dv;
        const int64_t hd = (int64_t)a.H * HD;
        const bool packed = a.dq_sn == 3 * hd && a.dkv_sn == 3 * hd && a.Nq == a.Nk && dk == dq + hd && dv == dq + 2 * hd &&
                            a.amax_dq && a.amax_dq == a.amax_dk && a.amax_dq == a.amax_dv;
        const bool plain = (a.dq_sn == 0 || a.dq_sn == hd) && (a.dkv_sn == 0 || a.dkv_sn == hd);
        if (want && !packed && !plain) return VIT_EINVAL;
        if (rope) {
            hipLaunchKernelGGL(k_attn_bwd_kv<true>, gkv, dim3(256), 0, stream, b, q, k, v, dout, lse, delta_ws, dk, dv);
            hipLaunchKernelGGL(k_attn_bwd_q<true>, gq, dim3(256), 0, stream, b, q, k, v, dout, lse, delta_ws, dq);
        } else {
            hipLaunchKernelGGL(k_attn_bwd_kv<false>, gkv, dim3(256), 0, stream, b, q, k, v, dout, lse, delta_ws, dk, dv);
            hipLaunchKernelGGL(k_attn_bwd_q<false>, gq, dim3(256), 0, stream, b, q, k, v, dout, lse, delta_ws, dq);
        }
        e = hipGetLastError();
        if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
        if (q_tail || k_tail) { const int rc = attention_bwd_tails(b, q, k, v, lse, dout, delta_ws, dq, dk, dv, q_tail, k_tail, stream); if (rc != VIT_OK) return rc; }
        if (!want) return VIT_OK;
        if (packed) return amax(dq, (int64_t)a.B * a.Nq * 3 * hd, a.amax_dq, stream);
        int rc = VIT_OK;
        if (a.amax_dq) rc = amax(dq, (int64_t)a.B * a.Nq * hd, a.amax_dq, stream);
        if (rc == VIT_OK && a.amax_dk) rc = amax(dk, (int64_t)a.B * a.Nk * hd, a.amax_dk, stream);
        if (rc == VIT_OK && a.amax_dv) rc = amax(dv, (int64_t)a.B * a.Nk * hd, a.amax_dv, stream);
        return rc;
    }
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    if (q_tail || k_tail) return attention_bwd_tails(a, q, k, v, lse, dout, delta_ws, dq, dk, dv, q_tail, k_tail, stream);
    return VIT_OK;
}
}  // namespace vit
