// vit_attention_x6.hip -- flash attention forward at fp32 accuracy on the bf16 matrix cores (head_dim 64, no mask).
//
// vit_attention.hip runs both contractions on the exact-f32 MFMA (157 TF peak, 64 cycles per 32x32x2 step).  Here Q, K, V
// and the probabilities are split into three bf16 pieces (vit_gemm_x6.hip has the arithmetic) and every contraction
// takes six v_mfma_f32_32x32x16_bf16 per 16-wide step: 24 MFMAs x 32 cycles instead of 32 x 64 per (32 x 32 x 64) block,
// 2.7x less matrix-pipe time, and the pipes that are busy are the bf16 ones the power budget is kinder to.
//
// Same orientation as the f32 kernel -- the query is always the MFMA column = the lane:
//     S^T (32 keys x 32 queries) = K (32 x 64) . Q^T          A = K pieces from LDS,   B = Q pieces in registers
//     O^T (32 d    x 32 queries) = V^T (32 x keys) . P^T      A = V^T pieces from LDS, B = P pieces in registers
// The D layout of S^T gives a lane the keys (r & 3) + 8 (r >> 2) + 4 half of a 32-key block; a contraction does not
// care about the order of its terms, so the second product simply DEFINES k-slot (step u, half, j) as key
// (j & 3) + 8 (2u + (j >> 2)) + 4 half: the probabilities a lane holds after the softmax are, after the split, its
// B fragments (registers 8u .. 8u+7 -> step u), no cross-lane movement, and V^T is stored in LDS with its key axis
// permuted the same way (position 16u + 8 half + j), so an A fragment is one 16-byte read.
//
// LDS per workgroup (128 queries, 4 wavefronts; 64-key tiles): K pieces [key][8-wide d group][piece][8 bf16] with a
// 400-byte row (25 16-byte slots: 16 consecutive rows hit 16 distinct 4-bank slots), V^T pieces [piece][d][64 permuted
// keys] with a 144-byte row (9 slots, same property): 25.6 + 27.6 KB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"
#include "vit_amax.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

namespace ax6 {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HD = 64, QW = 32, QB = 128, KT = 64;
constexpr int KROW = 400;            // bytes per key row of the K image
constexpr int VROW = 144;            // bytes per d row of one V^T piece plane
constexpr int K_BYTES = KT * KROW, V_BYTES = 3 * HD * VROW;

__device__ inline void split2(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    f32x2 f = {a, b};
    const bf16x2 h0 = __builtin_convertvector(f, bf16x2);
    const f32x2 r1 = f - __builtin_convertvector(h0, f32x2);
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    p0 = __builtin_bit_cast(uint32_t, h0); p1 = __builtin_bit_cast(uint32_t, h1); p2 = __builtin_bit_cast(uint32_t, h2);
}
// eight fp32 values -> three bf16x8 pieces
__device__ inline void split8(const float *v, bf16x8 &f0, bf16x8 &f1, bf16x8 &f2)
{
    uint4 q0, q1, q2;
    split2(v[0], v[1], q0.x, q1.x, q2.x);
    split2(v[2], v[3], q0.y, q1.y, q2.y);
    split2(v[4], v[5], q0.z, q1.z, q2.z);
    split2(v[6], v[7], q0.w, q1.w, q2.w);
    f0 = __builtin_bit_cast(bf16x8, q0); f1 = __builtin_bit_cast(bf16x8, q1); f2 = __builtin_bit_cast(bf16x8, q2);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NP == 2 ("f16x3"): eight ALREADY SCALED fp32 values -> two fp16x8 pieces (in bf16x8 registers: only the MFMA reinterprets them)
__device__ inline void split8h(const float *v, bf16x8 &f0, bf16x8 &f1)
{
    uint4 q0, q1;
    f16_split2(v[0], v[1], q0.x, q1.x);
    f16_split2(v[2], v[3], q0.y, q1.y);
    f16_split2(v[4], v[5], q0.z, q1.z);
    f16_split2(v[6], v[7], q0.w, q1.w);
    f0 = __builtin_bit_cast(bf16x8, q0); f1 = __builtin_bit_cast(bf16x8, q1);
}
template <int NP> __device__ inline void split8p(const float *v, bf16x8 &f0, bf16x8 &f1, bf16x8 &f2)
{
    if (NP == 2) split8h(v, f0, f1); else split8(v, f0, f1, f2);
}
__device__ inline float wave_xor32(float x)
{
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return (threadIdx.x & 32) ? x : y;
}

// NP = 6: six partial products, smallest first; NP = 3 ("bf16x3"): without the three 2^-16-level products (see vit_attention_bwd_x6.hip);
// NP = 2 ("f16x3", round 6): two fp16 pieces per operand, h l' + l h' + h h' on v_mfma_f32_32x32x16_f16 -- 2^-22 per product at the MFMA
// count of bf16x3.  Operands carry power-of-two scales: Q, K, V from their tensors' |max| words (VitAttnArgs.amax_q / _k / _v), the
// probabilities the constant 2^14 (p <= 1); the scores are un-scaled before the softmax, the output in the epilogue.
template <int NP>
__device__ inline f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c)
{
    if (NP == 2) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
        return c;
    }
    if (NP == 6) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

template <bool ROPE, int NP>
__global__ void __launch_bounds__(256, 2) k_attn_fwd_x6(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, float *__restrict__ out, float *__restrict__ lse)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_k[K_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char s_v[V_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * QB + wave * QW;
    const int qi = min(q0 + col, a.Nq - 1);   // clamped: rows beyond Nq compute garbage that is never stored
    const bool wave_active = q0 < a.Nq;
    float qscale = a.scale * 1.4426950408889634f;   // scores in the base-2 domain
    // f16x3: tensor scales (a rotation grows a component by at most sqrt 2: |max| s < 2^15.5, inside fp16's 65 504) and the two inverse factors
    float sk = 1.f, sv = 1.f, inv_qk = 1.f, inv_pv = 1.f;
    constexpr float PSCALE = 16384.f;
    if (NP == 2) {
        const float sq = f16_scale_of(amax_word_read(a.amax_q));
        sk = f16_scale_of(amax_word_read(a.amax_k)); sv = f16_scale_of(amax_word_read(a.amax_v));
        qscale *= sq;
        inv_qk = 1.f / (sq * sk); inv_pv = 1.f / (PSCALE * sv);
    }

    // ---- Q fragments: step t covers d = 16 t + 8 half + j, j = 0..7; rotated, pre-scaled, split once ----
    bf16x8 qf[4][3];
    {
        const float *qr = q + (int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh;
        float x[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 lo = *reinterpret_cast<const float4 *>(qr + 16 * t + 8 * half), hi = *reinterpret_cast<const float4 *>(qr + 16 * t + 8 * half + 4);
            x[t][0] = lo.x; x[t][1] = lo.y; x[t][2] = lo.z; x[t][3] = lo.w; x[t][4] = hi.x; x[t][5] = hi.y; x[t][6] = hi.z; x[t][7] = hi.w;
        }
        if (ROPE) {
            const int64_t py = a.qpos[((int64_t)b * a.Nq + qi) * 2 + 0], px = a.qpos[((int64_t)b * a.Nq + qi) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * half + j;     // 0..15: pairs (d, d+16) of [0,32) rotate by py, of [32,64) by px
                const float cy = a.cos_tab[py * 16 + d], sy = a.sin_tab[py * 16 + d];
                const float cx = a.cos_tab[px * 16 + d], sx = a.sin_tab[px * 16 + d];
                const float uy = x[0][j], vy = x[1][j], ux = x[2][j], vx = x[3][j];
                x[0][j] = uy * cy - vy * sy; x[1][j] = vy * cy + uy * sy;
                x[2][j] = ux * cx - vx * sx; x[3][j] = vx * cx + ux * sx;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[t][j] *= qscale;
            split8p<NP>(x[t], qf[t][0], qf[t][1], qf[t][2]);
        }
    }

    f32x16 o0 = {0}, o1 = {0};          // O^T rows d = rowmap(r) and 32 + rowmap(r), column = this lane's query
    float m = -INFINITY, l = 0.f;

    const float *kb = k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const float *vb = v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;

    // staging.  K: thread = (key, sub): d groups g = 4 (sub >> 1) + (sub & 1) and g + 2 (the RoPE partner, d + 16).
    // V: thread item = (4 consecutive keys, one d), four items per thread: the four keys are adjacent in the permuted key order too.
    const int skey = tid >> 2, sub = tid & 3, sg = 4 * (sub >> 1) + (sub & 1);
    float4 kreg[4];
    float vreg[4][4];
    int kpy = 0, kpx = 0;
    auto fetch = [&](int k0) {
        const int kg = min(k0 + skey, a.Nk - 1);
        const float *kr = kb + (int64_t)kg * a.k_sn + 8 * sg;
        kreg[0] = *reinterpret_cast<const float4 *>(kr); kreg[1] = *reinterpret_cast<const float4 *>(kr + 4);
        kreg[2] = *reinterpret_cast<const float4 *>(kr + 16); kreg[3] = *reinterpret_cast<const float4 *>(kr + 20);
        if (ROPE) { kpy = (int)a.kpos[((int64_t)b * a.Nk + kg) * 2 + 0]; kpx = (int)a.kpos[((int64_t)b * a.Nk + kg) * 2 + 1]; }
#pragma unroll
        for (int it = 0; it < 4; ++it) {       // item = (four consecutive keys 4 m .. 4 m + 3, one d): a lane walks d, so every load is a coalesced row piece
            const int m_ = wave + 4 * it;
#pragma unroll
            for (int e = 0; e < 4; ++e) vreg[it][e] = vb[(int64_t)min(k0 + 4 * m_ + e, a.Nk - 1) * a.v_sn + lane];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < a.Nk; k0 += KT) {
        __syncthreads();   // previous tile fully consumed
        {   // ---- K -> pieces, row-major ----
            const bool real = k0 + skey < a.Nk;     // rows beyond Nk are zeros (and masked to -inf below)
            float u[8] = {kreg[0].x, kreg[0].y, kreg[0].z, kreg[0].w, kreg[1].x, kreg[1].y, kreg[1].z, kreg[1].w};
            float w[8] = {kreg[2].x, kreg[2].y, kreg[2].z, kreg[2].w, kreg[3].x, kreg[3].y, kreg[3].z, kreg[3].w};
            if (ROPE) {
                const int pos = (sub >> 1) ? kpx : kpy;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d = 8 * (sub & 1) + j;
                    const float c = a.cos_tab[pos * 16 + d], s = a.sin_tab[pos * 16 + d];
                    const float t0 = u[j] * c - w[j] * s, t1 = w[j] * c + u[j] * s;
                    u[j] = t0; w[j] = t1;
                }
            }
            if (!real) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { u[j] = 0.f; w[j] = 0.f; }
            }
            if (NP == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { u[j] *= sk; w[j] *= sk; }
            }
            bf16x8 f0, f1, f2;
            bf16x8 *row = reinterpret_cast<bf16x8 *>(s_k + skey * KROW);
            split8p<NP>(u, f0, f1, f2);
            row[sg * 3 + 0] = f0; row[sg * 3 + 1] = f1;
            if (NP == 6) row[sg * 3 + 2] = f2;
            split8p<NP>(w, f0, f1, f2);
            row[(sg + 2) * 3 + 0] = f0; row[(sg + 2) * 3 + 1] = f1;
            if (NP == 6) row[(sg + 2) * 3 + 2] = f2;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {   // ---- V -> transposed pieces, key axis permuted ----
            // keys 4 m + e: key = 32 blk + (j & 3) + 8 (2u + (j >> 2)) + 4 hh  ->  position 32 blk + 16 u + 8 hh + j, j = e + 4 ((m >> 1) & 1)
            const int m_ = wave + 4 * it;
            const int pos = 32 * (m_ >> 3) + 16 * ((m_ >> 2) & 1) + 8 * (m_ & 1) + 4 * ((m_ >> 1) & 1);
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (k0 + 4 * m_ + e < a.Nk) ? vreg[it][e] : 0.f;
            uint2 w0, w1, w2;                            // packed pairs: low half = first value
            if (NP == 2) {
                f16_split2(x[0] * sv, x[1] * sv, w0.x, w1.x);
                f16_split2(x[2] * sv, x[3] * sv, w0.y, w1.y);
            } else {
                split2(x[0], x[1], w0.x, w1.x, w2.x);
                split2(x[2], x[3], w0.y, w1.y, w2.y);
            }
            unsigned char *dst = s_v + lane * VROW + pos * 2;
            *reinterpret_cast<uint2 *>(dst) = w0;
            *reinterpret_cast<uint2 *>(dst + HD * VROW) = w1;
            if (NP == 6) *reinterpret_cast<uint2 *>(dst + 2 * HD * VROW) = w2;
        }
        __syncthreads();
        if (k0 + KT < a.Nk) fetch(k0 + KT);
        if (!wave_active) continue;
        const bool two = k0 + 32 < a.Nk;   // second 32-key block holds at least one real key

        // ---- S^T = K Q^T for the two 32-key blocks ----
        f32x16 st0 = {0}, st1 = {0};
        {
            const unsigned char *ka = s_k + col * KROW + half * 48;           // key = col, d group 2t + half
            const unsigned char *kc = s_k + (32 + col) * KROW + half * 48;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 kf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const bf16x8 *>(ka + t * 96 + p * 16);
                st0 = mfma6<NP>(kf, qf[t], st0);
                if (two) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const bf16x8 *>(kc + t * 96 + p * 16);
                    st1 = mfma6<NP>(kf, qf[t], st1);
                }
            }
        }
        mfma_result_fence();            // (the branch around the second block's MFMAs joins here: see vit_amax.h)
        if (NP == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st0[r] *= inv_qk; st1[r] *= inv_qk; }
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

        // ---- O^T += V^T P^T ----
        {
            const unsigned char *va = s_v + col * VROW + half * 16;            // d = col (+ 32), keys at position 16 u + 8 half (+ 32 blk)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                if (blk == 1 && !two) break;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = (blk ? st1[8 * u + j] : st0[8 * u + j]) * (NP == 2 ? PSCALE : 1.f);
                    bf16x8 pf[3], vf[3];
                    split8p<NP>(pv, pf[0], pf[1], pf[2]);
                    const unsigned char *vp = va + blk * 64 + u * 32;
#pragma unroll
                    for (int p = 0; p < 3; ++p) vf[p] = *reinterpret_cast<const bf16x8 *>(vp + p * HD * VROW);
                    o0 = mfma6<NP>(vf, pf, o0);
#pragma unroll
                    for (int p = 0; p < 3; ++p) vf[p] = *reinterpret_cast<const bf16x8 *>(vp + 32 * VROW + p * HD * VROW);
                    o1 = mfma6<NP>(vf, pf, o1);
                }
            }
        }
    }

    // ---- epilogue: O = O^T / l, out[q][d], d = 8g + 4 half + {0..3} (+32) ----
    mfma_result_fence();                // (loop exit right behind the last P V MFMAs)
    uint32_t omax = 0;              // |max| of the stored values (VitAttnArgs.amax_out)
    if (q0 + col < a.Nq) {
        const float inv = (NP == 2 ? inv_pv : 1.f) / l;
        if (a.amax_out) {
#pragma unroll
            for (int r = 0; r < 16; ++r) omax = max(omax, max(abs_bits(o0[r] * inv), abs_bits(o1[r] * inv)));
        }
        float *orow = out + (int64_t)b * a.o_sb + (int64_t)(q0 + col) * a.o_sn + (int64_t)h * a.o_sh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * half;
            *reinterpret_cast<float4 *>(orow + d) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + d) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
        if (lse && half == 0) lse[((int64_t)b * a.H + h) * a.Nq + q0 + col] = (m + log2f(l)) * 0.6931471805599453f;
    }
    if (a.amax_out) amax_word_fold(a.amax_out, omax);      // (wave-uniform branch; waves past Nq fold 0)
}
}  // namespace ax6

// launched by attention_fwd (vit_attention.hip) when the split-arithmetic mode is on; same grid, same tail handling
hipError_t launch_attention_fwd_x6(const VitAttnArgs &a, const float *q, const float *k, const float *v, float *out, float *lse, dim3 grid,
                                   int products, hipStream_t stream)
{
    if (a.cos_tab) {
        if (products == 2) hipLaunchKernelGGL((ax6::k_attn_fwd_x6<true, 2>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
        else if (products == 3) hipLaunchKernelGGL((ax6::k_attn_fwd_x6<true, 3>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
        else hipLaunchKernelGGL((ax6::k_attn_fwd_x6<true, 6>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
    } else {
        if (products == 2) hipLaunchKernelGGL((ax6::k_attn_fwd_x6<false, 2>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
        else if (products == 3) hipLaunchKernelGGL((ax6::k_attn_fwd_x6<false, 3>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
        else hipLaunchKernelGGL((ax6::k_attn_fwd_x6<false, 6>), grid, dim3(256), 0, stream, a, q, k, v, out, lse);
    }
    return hipGetLastError();
}
}  // namespace vit
