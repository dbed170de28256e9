// vit_attention_bwd_x6.hip -- flash attention backward at fp32 accuracy on the bf16 matrix cores (head_dim 64, no mask).
//
// The same two passes as vit_attention_bwd.hip (k_attn_delta is shared), with every contraction in bf16x6 split arithmetic
// (six v_mfma_f32_32x32x16_bf16 per 16-wide step; vit_gemm_x6.hip has the arithmetic, vit_attention_x6.hip the forward):
//   k_attn_bwd_q_x6    a wavefront owns 32 QUERIES (query = MFMA column = lane) and walks 32-key tiles:
//                        S^T  = K Q^T        A = K rows  (LDS)   B = Q  pieces (regs)
//                        dP^T = V dO^T       A = V rows  (LDS)   B = dO pieces (regs)
//                        dQ^T += K^T dS^T    A = K^T     (LDS)   B = dS pieces (regs, split as produced)
//   k_attn_bwd_kv_x6   a wavefront owns 32 KEYS (key = lane) and walks 32-query tiles:
//                        S  = Q K^T, dP = dO V^T          A = Q / dO rows (LDS)       B = K / V pieces (regs)
//                        dV^T += dO^T P, dK^T += Q^T dS   A = dO^T / Q^T  (LDS)       B = P / dS pieces (regs)
// As in the forward, the operand that comes out of the first products (P, dS) is consumed in the register order the MFMA D
// layout produced it: k-slot (step u, half, j) of the second products IS row (j & 3) + 8 (2u + (j >> 2)) + 4 half of the tile, and
// the transposed LDS images (K^T, dO^T, Q^T) are written with that axis permuted accordingly (position 16 u + 8 half + j), so
// nothing crosses lanes and every fragment is one 16-byte read.
//
// LDS images of a 32-row tile: "rows" = [row][8-wide d group][piece][8 bf16], 400-byte rows; "transposed" = [piece][d][32
// permuted rows] with an 80-byte row (5 16-byte slots).  Both strides put 16 consecutive rows on 16 distinct 4-bank slots.
// With fused RoPE, Q and K are rotated while they are staged / loaded and dQ, dK are rotated back before they are stored
// (lane-local: features d and d + 16 sit in registers r and r + 8 of the same accumulator).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"
#include "vit_amax.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

namespace abx6 {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HD = 64, TR = 32;                 // head dim, rows per tile
constexpr int ROWB = 400, TROWB = 80;           // bytes per row of the two image kinds
constexpr int ROWS_BYTES = TR * ROWB, T_BYTES = 3 * HD * TROWB;
constexpr float LOG2E = 1.4426950408889634f;

__device__ inline int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

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
__device__ inline void split8(const float *v, bf16x8 (&f)[3])
{
    uint4 q0, q1, q2;
    split2(v[0], v[1], q0.x, q1.x, q2.x);
    split2(v[2], v[3], q0.y, q1.y, q2.y);
    split2(v[4], v[5], q0.z, q1.z, q2.z);
    split2(v[6], v[7], q0.w, q1.w, q2.w);
    f[0] = __builtin_bit_cast(bf16x8, q0); f[1] = __builtin_bit_cast(bf16x8, q1); f[2] = __builtin_bit_cast(bf16x8, q2);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NP == 2 ("f16x3", vit_attention_set_arith(3), round 6): eight ALREADY SCALED values -> two fp16x8 pieces (held in bf16x8 registers)
__device__ inline void split8h(const float *v, bf16x8 (&f)[3])
{
    uint4 q0, q1;
    f16_split2(v[0], v[1], q0.x, q1.x);
    f16_split2(v[2], v[3], q0.y, q1.y);
    f16_split2(v[4], v[5], q0.z, q1.z);
    f16_split2(v[6], v[7], q0.w, q1.w);
    f[0] = __builtin_bit_cast(bf16x8, q0); f[1] = __builtin_bit_cast(bf16x8, q1);
}
template <int NP> __device__ inline void split8p(const float *v, bf16x8 (&f)[3]) { if (NP == 2) split8h(v, f); else split8(v, f); }
template <int NP> __device__ inline void split2p(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    if (NP == 2) f16_split2(a, b, p0, p1); else split2(a, b, p0, p1, p2);
}
// NP = 6: six partial products, smallest first; NP = 3 ("bf16x3", vit_attention_set_arith(2)): the three 2^-16-level products are left
// out -- the third bf16 piece of every operand is then never used: the compiler drops its computation, the image stores skip it;
// NP = 2 ("f16x3"): two fp16 pieces of value x power-of-two scale, h l' + l h' + h h' on the f16 MFMA (2^-22 per product).  Scales: Q, K, V, dO
// from their tensors' |max| words (folded into `mul` of the loaders / stagers below), P the constant 2^14, dS a PER-LANE RUNNING scale (the
// lane is the MFMA column = the query / key that owns the accumulators, so a lane's scale multiplies its whole accumulator column: when a
// tile's dS outgrows the lane's current scale the column is rescaled by an exact power of two, as the online softmax rescales O).
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

// ---- register fragments of one row (the MFMA B operand): step t covers d = 16 t + 8 half + j; rotated if ROPE, times `mul` ----
template <bool ROPE, int NP>
__device__ inline void load_row_pieces(bf16x8 (&f)[4][3], const float *__restrict__ rp, int half, const int64_t *__restrict__ pos2,
                                       const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, float mul)
{
    float x[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 lo = *reinterpret_cast<const float4 *>(rp + 16 * t + 8 * half), hi = *reinterpret_cast<const float4 *>(rp + 16 * t + 8 * half + 4);
        x[t][0] = lo.x; x[t][1] = lo.y; x[t][2] = lo.z; x[t][3] = lo.w; x[t][4] = hi.x; x[t][5] = hi.y; x[t][6] = hi.z; x[t][7] = hi.w;
    }
    if (ROPE) {
        const int64_t py = pos2[0], px = pos2[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * half + j;     // 0..15: pairs (d, d+16) of [0,32) rotate by py, of [32,64) by px
            const float cy = cos_tab[py * 16 + d], sy = sin_tab[py * 16 + d];
            const float cx = cos_tab[px * 16 + d], sx = sin_tab[px * 16 + d];
            const float uy = x[0][j], vy = x[1][j], ux = x[2][j], vx = x[3][j];
            x[0][j] = uy * cy - vy * sy; x[1][j] = vy * cy + uy * sy;
            x[2][j] = ux * cx - vx * sx; x[3][j] = vx * cx + ux * sx;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[t][j] *= mul;
        split8p<NP>(x[t], f[t]);
    }
}

// ---- staging of a 32-row tile of a strided (b, n, h, 64) tensor into both image kinds ----
// Rows image: item (row = i >> 2, sub = i & 3), i in [0, 128): d groups sg = 4 (sub >> 1) + (sub & 1) and sg + 2 (the RoPE partner).
struct RowItem { float4 a0, a1, b0, b1; int py, px; };
template <bool ROPE>
__device__ inline void fetch_row_item(RowItem &it, const float *__restrict__ base, int64_t sn, int row0, int n_valid, int i,
                                      const int64_t *__restrict__ pos)
{
    const int row = i >> 2, sub = i & 3, sg = 4 * (sub >> 1) + (sub & 1);
    const int gi = min(row0 + row, n_valid - 1);
    const float *rp = base + (int64_t)gi * sn + 8 * sg;
    it.a0 = *reinterpret_cast<const float4 *>(rp); it.a1 = *reinterpret_cast<const float4 *>(rp + 4);
    it.b0 = *reinterpret_cast<const float4 *>(rp + 16); it.b1 = *reinterpret_cast<const float4 *>(rp + 20);
    if (ROPE) { it.py = (int)pos[(int64_t)gi * 2 + 0]; it.px = (int)pos[(int64_t)gi * 2 + 1]; }
}
template <bool ROPE, int NP>
__device__ inline void store_row_item(unsigned char *__restrict__ img, const RowItem &it, int row0, int n_valid, int i,
                                      const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, float mul = 1.f)
{
    const int row = i >> 2, sub = i & 3, sg = 4 * (sub >> 1) + (sub & 1);
    float u[8] = {it.a0.x, it.a0.y, it.a0.z, it.a0.w, it.a1.x, it.a1.y, it.a1.z, it.a1.w};
    float w[8] = {it.b0.x, it.b0.y, it.b0.z, it.b0.w, it.b1.x, it.b1.y, it.b1.z, it.b1.w};
    if (ROPE) {
        const int pos = (sub >> 1) ? it.px : it.py;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * (sub & 1) + j;
            const float c = cos_tab[pos * 16 + d], s = sin_tab[pos * 16 + d];
            const float t0 = u[j] * c - w[j] * s, t1 = w[j] * c + u[j] * s;
            u[j] = t0; w[j] = t1;
        }
    }
    if (row0 + row >= n_valid) {       // rows past the end are zeros
#pragma unroll
        for (int j = 0; j < 8; ++j) { u[j] = 0.f; w[j] = 0.f; }
    }
    if (NP == 2) {                     // f16x3: the tensor's power-of-two scale
#pragma unroll
        for (int j = 0; j < 8; ++j) { u[j] *= mul; w[j] *= mul; }
    }
    bf16x8 f[3];
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(img + row * ROWB);
    split8p<NP>(u, f);
    dst[sg * 3 + 0] = f[0]; dst[sg * 3 + 1] = f[1];
    if (NP == 6) dst[sg * 3 + 2] = f[2];
    split8p<NP>(w, f);
    dst[(sg + 2) * 3 + 0] = f[0]; dst[(sg + 2) * 3 + 1] = f[1];
    if (NP == 6) dst[(sg + 2) * 3 + 2] = f[2];
}

// Transposed image: item (m = i >> 5: rows 4 m .. 4 m + 3, pd = i & 31: the feature pair (d, d + 16), d = pd + 16 (pd >> 4)),
// i in [0, 256).  Rows 4 m + e are adjacent in the permuted order too: positions p0 + e.
struct TItem { float u[4], w[4]; int pos[4]; };
template <bool ROPE>
__device__ inline void fetch_t_item(TItem &it, const float *__restrict__ base, int64_t sn, int row0, int n_valid, int i,
                                    const int64_t *__restrict__ pos)
{
    const int m = i >> 5, pd = i & 31, d = pd + 16 * (pd >> 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gi = min(row0 + 4 * m + e, n_valid - 1);
        const float *rp = base + (int64_t)gi * sn;
        it.u[e] = rp[d]; it.w[e] = rp[d + 16];
        if (ROPE) it.pos[e] = (int)pos[(int64_t)gi * 2 + (pd >> 4)];
    }
}
template <bool ROPE, int NP>
__device__ inline void store_t_item(unsigned char *__restrict__ img, const TItem &it, int row0, int n_valid, int i,
                                    const float *__restrict__ cos_tab, const float *__restrict__ sin_tab, float mul = 1.f)
{
    const int m = i >> 5, pd = i & 31, d = pd + 16 * (pd >> 4);
    const int p0 = 16 * ((m >> 2) & 1) + 8 * (m & 1) + 4 * ((m >> 1) & 1);
    float u[4], w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = it.u[e]; w[e] = it.w[e];
        if (ROPE) {
            const float c = cos_tab[it.pos[e] * 16 + (pd & 15)], s = sin_tab[it.pos[e] * 16 + (pd & 15)];
            const float t0 = u[e] * c - w[e] * s, t1 = w[e] * c + u[e] * s;
            u[e] = t0; w[e] = t1;
        }
        if (row0 + 4 * m + e >= n_valid) { u[e] = 0.f; w[e] = 0.f; }
        if (NP == 2) { u[e] *= mul; w[e] *= mul; }
    }
    uint2 a0, a1, a2;
    split2p<NP>(u[0], u[1], a0.x, a1.x, a2.x);
    split2p<NP>(u[2], u[3], a0.y, a1.y, a2.y);
    unsigned char *dst = img + d * TROWB + p0 * 2;
    *reinterpret_cast<uint2 *>(dst) = a0; *reinterpret_cast<uint2 *>(dst + HD * TROWB) = a1;
    if (NP == 6) *reinterpret_cast<uint2 *>(dst + 2 * HD * TROWB) = a2;
    split2p<NP>(w[0], w[1], a0.x, a1.x, a2.x);
    split2p<NP>(w[2], w[3], a0.y, a1.y, a2.y);
    dst += 16 * TROWB;
    *reinterpret_cast<uint2 *>(dst) = a0; *reinterpret_cast<uint2 *>(dst + HD * TROWB) = a1;
    if (NP == 6) *reinterpret_cast<uint2 *>(dst + 2 * HD * TROWB) = a2;
}

// first products of a tile: acc (32 tile rows x 32 lanes) = rows image . register pieces^T
template <int NP>
__device__ inline f32x16 rows_times_regs(const unsigned char *__restrict__ img, int col, int half, const bf16x8 (&reg)[4][3])
{
    f32x16 acc = {0};
    const unsigned char *ra = img + col * ROWB + half * 48;           // tile row = col, d group 2t + half
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bf16x8 f[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const bf16x8 *>(ra + t * 96 + p * 16);
        acc = mfma6<NP>(f, reg[t], acc);
    }
    return acc;
}
// second products: (lo, hi) (64 d x 32 lanes) += transposed image . x, x = 16 values per lane in D-layout register order
// (xmul: f16x3 only -- the scale of the register operand, 2^14 for P, the lane's running scale for dS)
template <int NP>
__device__ inline void t_times_regs(const unsigned char *__restrict__ img, int col, int half, const f32x16 &x, f32x16 &lo, f32x16 &hi, float xmul = 1.f)
{
    const unsigned char *ta = img + col * TROWB + half * 16;           // d = col (+ 32), positions 16 u + 8 half ..
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = NP == 2 ? x[8 * u + j] * xmul : x[8 * u + j];
        bf16x8 xf[3], tf[3];
        split8p<NP>(xv, xf);
#pragma unroll
        for (int p = 0; p < 3; ++p) tf[p] = *reinterpret_cast<const bf16x8 *>(ta + u * 32 + p * HD * TROWB);
        lo = mfma6<NP>(tf, xf, lo);
#pragma unroll
        for (int p = 0; p < 3; ++p) tf[p] = *reinterpret_cast<const bf16x8 *>(ta + u * 32 + 32 * TROWB + p * HD * TROWB);
        hi = mfma6<NP>(tf, xf, hi);
    }
}

// f16x3: the per-lane running scale of the dS operand.  `ds`: this lane's 16 dS values of the tile (its MFMA column: lanes l and l + 32 hold the
// two halves of the same column).  Returns the scale to split them with; when the column's |max| has outgrown the scale so far, the lane's
// accumulators (lo, hi: everything accumulated under the old scale) are multiplied by new / old first -- exact, powers of two.  The scale only
// ever shrinks (a larger |max| wins), so a lane rescales a handful of times at the start of its walk and then never again.
__device__ inline float ds_running_scale(const f32x16 &ds, float &s_run, f32x16 &lo, f32x16 &hi)
{
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = max(m, abs_bits(ds[r]));
    m = max(m, (uint32_t)__shfl_xor((int)m, 32, 64));                 // the column's other half
    const float s_new = fminf(s_run, f16_scale_of(m));                // f16_scale_of(0) = 1: an all-zero tile changes nothing below 1 ...
    if (m != 0u && s_new != s_run) {                                  // (per lane; rare)
        const float f = s_new / s_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { lo[r] *= f; hi[r] *= f; }
        s_run = s_new;
    }
    return s_run;
}

// inverse rotation of a transposed 64 x (lane) gradient held as two f32x16 (rows rowmap(r) and 32 + rowmap(r))
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
__device__ inline uint32_t max_abs_32(const f32x16 &lo, const f32x16 &hi)
{
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = max(m, max(abs_bits(lo[r]), abs_bits(hi[r])));
    return m;
}
__device__ inline void store_64(float *__restrict__ row, const f32x16 &lo, const f32x16 &hi, int half)
{
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int d = 8 * gq + 4 * half;
        *reinterpret_cast<float4 *>(row + d) = make_float4(lo[4 * gq], lo[4 * gq + 1], lo[4 * gq + 2], lo[4 * gq + 3]);
        *reinterpret_cast<float4 *>(row + 32 + d) = make_float4(hi[4 * gq], hi[4 * gq + 1], hi[4 * gq + 2], hi[4 * gq + 3]);
    }
}

// ------------------------------------------------------------------ dQ
template <bool ROPE, int NP>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_q_x6(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                          const float *__restrict__ v, const float *__restrict__ g,
                                                          const float *__restrict__ lse, const float *__restrict__ delta,
                                                          float *__restrict__ dq)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_k[ROWS_BYTES], s_v[ROWS_BYTES], s_kt[T_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = min(q0 + col, a.Nq - 1);
    const bool wave_active = q0 < a.Nq;

    // f16x3: power-of-two operand scales from the tensors' |max| words; the products are un-scaled right behind their MFMAs
    float sk = 1.f, sv = 1.f, sg = 1.f, sq = 1.f, inv_qk = 1.f, inv_gv = 1.f, s_run = 1.2676506e30f /* 2^100: no dS seen yet */;
    if (NP == 2) {
        sq = f16_scale_of(amax_word_read(a.amax_q)); sk = f16_scale_of(amax_word_read(a.amax_k));
        sv = f16_scale_of(amax_word_read(a.amax_v)); sg = f16_scale_of(amax_word_read(a.amax_g));
        inv_qk = 1.f / (sq * sk); inv_gv = 1.f / (sg * sv);
    }
    bf16x8 qf[4][3], gf[4][3];      // Q pre-scaled by scale * log2(e): only S consumes it
    load_row_pieces<ROPE, NP>(qf, q + (int64_t)b * a.q_sb + (int64_t)qi * a.q_sn + (int64_t)h * a.q_sh, half,
                              ROPE ? a.qpos + ((int64_t)b * a.Nq + qi) * 2 : nullptr, a.cos_tab, a.sin_tab, a.scale * LOG2E * sq);
    load_row_pieces<false, NP>(gf, g + (((int64_t)b * a.Nq + qi) * a.H + h) * HD, half, nullptr, nullptr, nullptr, sg);
    const float lse2 = lse[((int64_t)b * a.H + h) * a.Nq + qi] * LOG2E;
    const float del = delta[((int64_t)b * a.H + h) * a.Nq + qi];

    f32x16 dq0 = {0}, dq1 = {0};
    const float *kb = k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const float *vb = v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int64_t *kpos = ROPE ? a.kpos + (int64_t)b * a.Nk * 2 : nullptr;

    // threads 0..127 stage the K rows, 128..255 the V rows; every thread one item of K^T
    RowItem ri; TItem ti;
    auto fetch = [&](int k0) {
        if (tid < 128) fetch_row_item<ROPE>(ri, kb, a.k_sn, k0, a.Nk, tid, kpos);
        else fetch_row_item<false>(ri, vb, a.v_sn, k0, a.Nk, tid - 128, nullptr);
        fetch_t_item<ROPE>(ti, kb, a.k_sn, k0, a.Nk, tid, kpos);
    };
    fetch(0);
    for (int k0 = 0; k0 < a.Nk; k0 += TR) {
        __syncthreads();
        if (tid < 128) store_row_item<ROPE, NP>(s_k, ri, k0, a.Nk, tid, a.cos_tab, a.sin_tab, sk);
        else store_row_item<false, NP>(s_v, ri, k0, a.Nk, tid - 128, nullptr, nullptr, sv);
        store_t_item<ROPE, NP>(s_kt, ti, k0, a.Nk, tid, a.cos_tab, a.sin_tab, sk);
        __syncthreads();
        if (k0 + TR < a.Nk) fetch(k0 + TR);
        if (!wave_active) continue;
        f32x16 st = rows_times_regs<NP>(s_k, col, half, qf);
        f32x16 dp = rows_times_regs<NP>(s_v, col, half, gf);
        if (NP == 2) mfma_result_fence();
        // element r: key k0 + rowmap(r), query = this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + rowmap(r, half);
            const float sr = NP == 2 ? st[r] * inv_qk : st[r], dr = NP == 2 ? dp[r] * inv_gv : dp[r];
            const float p = key < a.Nk ? exp2f(sr - lse2) : 0.f;
            dp[r] = p * (dr - del) * a.scale;
        }
        const float sds = NP == 2 ? ds_running_scale(dp, s_run, dq0, dq1) : 1.f;
        t_times_regs<NP>(s_kt, col, half, dp, dq0, dq1, sds);
    }
    mfma_result_fence();    // (loop exit right behind the last MFMAs: vit_amax.h)
    if (NP == 2) {          // accumulated under (the lane's final dS scale) x (K's scale)
        const float f = 1.f / (s_run * sk);
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq0[r] *= f; dq1[r] *= f; }
    }
    if (q0 + col < a.Nq) {
        if (ROPE) {
            const int64_t *pp = a.qpos + ((int64_t)b * a.Nq + q0 + col) * 2;
            unrotate(dq0, dq1, half, pp[0], pp[1], a.cos_tab, a.sin_tab);
        }
        store_64(dq + ((int64_t)b * a.Nq + q0 + col) * (a.dq_sn ? a.dq_sn : (int64_t)a.H * HD) + h * HD, dq0, dq1, half);   // (B,Nq,H,64), token stride dq_sn
    }
    if (a.amax_dq) amax_word_fold(a.amax_dq, (q0 + col < a.Nq) ? max_abs_32(dq0, dq1) : 0u);      // |max| of the stored gradient (after the inverse rotation)
}

// ------------------------------------------------------------------ dK, dV
template <bool ROPE, int NP>
__global__ void __launch_bounds__(256, 2) k_attn_bwd_kv_x6(VitAttnArgs a, const float *__restrict__ q, const float *__restrict__ k,
                                                           const float *__restrict__ v, const float *__restrict__ g,
                                                           const float *__restrict__ lse, const float *__restrict__ delta,
                                                           float *__restrict__ dk, float *__restrict__ dv)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_q[ROWS_BYTES], s_g[ROWS_BYTES], s_qt[T_BYTES], s_gt[T_BYTES];
    __shared__ float s_lse[TR], s_delta[TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int key0 = blockIdx.x * 128 + wave * 32;
    const int ki = min(key0 + col, a.Nk - 1);
    const bool wave_active = key0 < a.Nk;

    float sk = 1.f, sv = 1.f, sg = 1.f, sq = 1.f, inv_qk = 1.f, inv_gv = 1.f, s_run = 1.2676506e30f /* 2^100: no dS seen yet */;
    constexpr float PSCALE = 16384.f;       // f16x3: the probabilities' scale (p <= 1)
    if (NP == 2) {
        sq = f16_scale_of(amax_word_read(a.amax_q)); sk = f16_scale_of(amax_word_read(a.amax_k));
        sv = f16_scale_of(amax_word_read(a.amax_v)); sg = f16_scale_of(amax_word_read(a.amax_g));
        inv_qk = 1.f / (sq * sk); inv_gv = 1.f / (sg * sv);
    }
    bf16x8 kf[4][3], vf[4][3];      // K pre-scaled by scale * log2(e): only S consumes it
    load_row_pieces<ROPE, NP>(kf, k + (int64_t)b * a.k_sb + (int64_t)ki * a.k_sn + (int64_t)h * a.k_sh, half,
                              ROPE ? a.kpos + ((int64_t)b * a.Nk + ki) * 2 : nullptr, a.cos_tab, a.sin_tab, a.scale * LOG2E * sk);
    load_row_pieces<false, NP>(vf, v + (int64_t)b * a.v_sb + (int64_t)ki * a.v_sn + (int64_t)h * a.v_sh, half, nullptr, nullptr, nullptr, sv);

    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const float *qb = q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const float *gb = g + ((int64_t)b * a.Nq * a.H + h) * HD;          // contiguous (B,Nq,H,64)
    const int64_t g_sn = (int64_t)a.H * HD;
    const float *lse_b = lse + ((int64_t)b * a.H + h) * a.Nq;
    const float *del_b = delta + ((int64_t)b * a.H + h) * a.Nq;
    const int64_t *qpos = ROPE ? a.qpos + (int64_t)b * a.Nq * 2 : nullptr;

    // threads 0..127 stage the Q rows, 128..255 the dO rows; every thread one item of Q^T and one of dO^T
    // (only the row item is prefetched across the MFMA phase: the two transposed items on top of the 96 registers of K / V pieces
    // and the four accumulators spilled 49 registers; their loads are issued at the top of the tile and hit the L2 lines the row
    // items of the same tile just fetched)
    RowItem ri;
    auto fetch = [&](int q0) {
        if (tid < 128) fetch_row_item<ROPE>(ri, qb, a.q_sn, q0, a.Nq, tid, qpos);
        else fetch_row_item<false>(ri, gb, g_sn, q0, a.Nq, tid - 128, nullptr);
    };
    fetch(0);
    for (int q0 = 0; q0 < a.Nq; q0 += TR) {
        TItem tq, tg;
        fetch_t_item<ROPE>(tq, qb, a.q_sn, q0, a.Nq, tid, qpos);
        fetch_t_item<false>(tg, gb, g_sn, q0, a.Nq, tid, nullptr);
        __syncthreads();
        if (tid < 128) store_row_item<ROPE, NP>(s_q, ri, q0, a.Nq, tid, a.cos_tab, a.sin_tab, sq);
        else store_row_item<false, NP>(s_g, ri, q0, a.Nq, tid - 128, nullptr, nullptr, sg);
        store_t_item<ROPE, NP>(s_qt, tq, q0, a.Nq, tid, a.cos_tab, a.sin_tab, sq);
        store_t_item<false, NP>(s_gt, tg, q0, a.Nq, tid, nullptr, nullptr, sg);
        if (tid < TR) {
            const int qi = q0 + tid;
            s_lse[tid] = qi < a.Nq ? lse_b[qi] * LOG2E : INFINITY;   // padded queries: P = exp2(-inf) = 0
            s_delta[tid] = qi < a.Nq ? del_b[qi] : 0.f;
        }
        __syncthreads();
        if (q0 + TR < a.Nq) fetch(q0 + TR);
        if (!wave_active) continue;
        f32x16 sc = rows_times_regs<NP>(s_q, col, half, kf);
        f32x16 dp = rows_times_regs<NP>(s_g, col, half, vf);
        if (NP == 2) mfma_result_fence();
        // element r: query q0 + rowmap(r), key = this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = rowmap(r, half);
            const float sr = NP == 2 ? sc[r] * inv_qk : sc[r], dr = NP == 2 ? dp[r] * inv_gv : dp[r];
            const float p = exp2f(sr - s_lse[qr]);
            sc[r] = p;                                        // P
            dp[r] = p * (dr - s_delta[qr]) * a.scale;         // dS (w.r.t. the unscaled dot product)
        }
        t_times_regs<NP>(s_gt, col, half, sc, dv0, dv1, PSCALE);
        const float sds = NP == 2 ? ds_running_scale(dp, s_run, dk0, dk1) : 1.f;
        t_times_regs<NP>(s_qt, col, half, dp, dk0, dk1, sds);
    }
    mfma_result_fence();
    if (NP == 2) {
        const float fk = 1.f / (s_run * sq), fv = 1.f / (PSCALE * sg);
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk0[r] *= fk; dk1[r] *= fk; dv0[r] *= fv; dv1[r] *= fv; }
    }
    if (key0 + col < a.Nk) {
        if (ROPE) {
            const int64_t *pp = a.kpos + ((int64_t)b * a.Nk + key0 + col) * 2;
            unrotate(dk0, dk1, half, pp[0], pp[1], a.cos_tab, a.sin_tab);
        }
        // dk, dv contiguous (B,Nk,H,64)
        store_64(dk + ((int64_t)b * a.Nk + key0 + col) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD, dk0, dk1, half);
        store_64(dv + ((int64_t)b * a.Nk + key0 + col) * (a.dkv_sn ? a.dkv_sn : (int64_t)a.H * HD) + h * HD, dv0, dv1, half);
    }
    const bool stored = key0 + col < a.Nk;
    if (a.amax_dk) amax_word_fold(a.amax_dk, stored ? max_abs_32(dk0, dk1) : 0u);
    if (a.amax_dv) amax_word_fold(a.amax_dv, stored ? max_abs_32(dv0, dv1) : 0u);
}
}  // namespace abx6

// launched by attention_bwd (vit_attention_bwd.hip) in place of its two f32 kernels when the split-arithmetic mode is on
hipError_t launch_attention_bwd_x6(const VitAttnArgs &a, const float *q, const float *k, const float *v, const float *dout, const float *lse,
                                   const float *delta, float *dq, float *dk, float *dv, dim3 gkv, dim3 gq, int products, hipStream_t stream)
{
#define ABX_LAUNCH(RP, NP_)                                                                                                              \
    do {                                                                                                                                 \
        hipLaunchKernelGGL((abx6::k_attn_bwd_kv_x6<RP, NP_>), gkv, dim3(256), 0, stream, a, q, k, v, dout, lse, delta, dk, dv);          \
        hipLaunchKernelGGL((abx6::k_attn_bwd_q_x6<RP, NP_>), gq, dim3(256), 0, stream, a, q, k, v, dout, lse, delta, dq);                \
    } while (0)
    if (a.cos_tab) { if (products == 2) ABX_LAUNCH(true, 2); else if (products == 3) ABX_LAUNCH(true, 3); else ABX_LAUNCH(true, 6); }
    else { if (products == 2) ABX_LAUNCH(false, 2); else if (products == 3) ABX_LAUNCH(false, 3); else ABX_LAUNCH(false, 6); }
#undef ABX_LAUNCH
    return hipGetLastError();
}
}  // namespace vit
