// vit_gemm_x6.hip -- fp32-accurate Linear on the bf16 matrix cores of gfx950 ("bf16x6" split arithmetic).
//
// gfx950 has no TF32/xf32 path and its exact-f32 MFMA runs at 1/16 of the bf16 rate (157 TF vs 2.5 PF).  An fp32
// value splits EXACTLY into three bf16 pieces, a = a0 + a1 + a2 + O(2^-27 |a|) (each piece is the round-to-nearest
// bf16 of the running residual; every residual is exactly representable in fp32), and a product of two bf16 values is
// exact in fp32.  Keeping the six partial products whose weight is >= 2^-18 of the leading one,
//     a.b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)          (dropped: a1 b2, a2 b1, a2 b2 <= 2^-26 |a||b|)
// and accumulating them in the MFMA's fp32 accumulator reproduces an fp32 GEMM to fp32 rounding accuracy at 6 bf16
// MFMAs per k-step: 16/6 = 2.7x the f32-MFMA peak (417 TF).  tests/test_gpu_vit.py measures the error of this path
// and of the f32-MFMA path against an fp64 reference on the same inputs.
//
//     out (M,N) = [residual +] act( x (M,K) . w^T (N,K) + bias ),     w pre-split ONCE per optimizer step
//
// The weight is static inside a step, so it is split ahead of time (vit_split_weight) into the exact order the MFMA
// B-operand wants: packed[n][k/8][piece][8] bf16, 48 contiguous bytes per (row, 8-wide k group).  Activations are
// split while they are staged into LDS (v_cvt_pk_bf16_f32 + one subtract per piece).  The same kernel computes the
// input gradient dX = dY . W from the pre-split TRANSPOSED weight (vit_split_weight with transpose = 1).
//
// 128 x (64 TN) output tile per workgroup, 4 wavefronts (2x2) of 64 x 32TN sub-tiles, v_mfma_f32_32x32x16_bf16,
// K consumed 16 at a time through double-buffered, XOR-swizzled 96-byte LDS rows; two register stages keep the global
// loads of slab k+2 in flight while slab k feeds the MFMAs; three workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace x6 {
constexpr int BM = 128, BK = 16, ROWQ = 6;   // LDS row = 6 x 16-B slots (2 k-groups x 3 pieces), no padding
// slot s of row r lives at physical slot s ^ ((r >> 3) & 1): with 96-byte rows the 16 lanes of every ds_read_b128 lane
// group ({0-3,12-15,20-27}, {4-11,16-19,28-31}: MI355X_MICROARCH LDS table) then hit 16 distinct 4-bank slots, and the
// 8-lane groups of ds_write_b128 8 distinct ones; 48 KiB per workgroup -> three workgroups per CU
__device__ inline int swz(int row, int slot) { return slot ^ ((row >> 3) & 1); }
// Swizzle of the images filled by a TRANSPOSING loader (lane = tile row, weight-gradient kernels): there the 8 lanes that share a
// ds_write_b128 cycle are 8 consecutive rows, 96 bytes apart -- only four distinct 16-byte columns of the 32-bank write path, a 2-way
// conflict on every store (r03 PMC: a third of the LDS cycles of k_wgrad_x6).  Flipping the slot pair on row bit 2 as well separates rows
// r and r + 4; the ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}) still pair rows whose flags differ.
__device__ inline int swz_t(int row, int slot) { return slot ^ (((row >> 3) ^ (row >> 2)) & 1); }

// Workgroup id -> output tile.  (1) Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own
// L2: XCD x gets one CONTIGUOUS range of the tile sequence (exact partition for any tile count).  (2) The sequence
// itself walks the tile grid in groups of GM row-tiles, column by column, so the ~64 tiles an XCD has in flight form
// an ~8x8 block: every A row-panel and every B column-panel fetched into that L2 is reused ~8 times instead of the
// 24x / 2.7x of a row-major walk (the B panels do not fit the 4 MB L2 and were re-streamed over the fabric).
__device__ inline void tile_of_block(int bid, int tiles_m, int tiles_n, int &tm, int &tn)
{
    constexpr int GM = 8;
    const int ntiles = tiles_m * tiles_n, q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int pid = xcd * q + min(xcd, r) + local;
    const int per_group = GM * tiles_n;
    const int group = pid / per_group, first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_group = pid - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
}

// Zero fill of a split-contraction output.  (Not hipMemsetAsync: its graph node is not replayed faithfully by this runtime -- captured
// into a hipGraph, the second and later replays leave half of the words untouched -- so the serving graphs of styl3r_amd/graphs.py
// would accumulate into stale sums; a kernel node replays as launched.  tools/probes/graph_memset_replay.py is the reproducer.)
__global__ void __launch_bounds__(256) k_zero_words(uint32_t *__restrict__ p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        uint4 *p4 = reinterpret_cast<uint4 *>(p);
        const size_t n4 = n >> 2;
        for (size_t j = i; j < n4; j += stride) p4[j] = make_uint4(0, 0, 0, 0);
        for (size_t j = (n4 << 2) + i; j < n; j += stride) p[j] = 0;
    } else {
        for (; i < n; i += stride) p[i] = 0;
    }
}

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// d/dx of the exact GELU: Phi(x) + x phi(x)   (aten's GeluBackward, approximate = "none")
__device__ inline float gelu_grad_exact(float x)
{
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// two fp32 values -> their three bf16 pieces, each packed (lo = first value)
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

__device__ inline void split8(const float4 &lo, const float4 &hi, uint4 &q0, uint4 &q1, uint4 &q2)
{
    split2(lo.x, lo.y, q0.x, q1.x, q2.x);
    split2(lo.z, lo.w, q0.y, q1.y, q2.y);
    split2(hi.x, hi.y, q0.z, q1.z, q2.z);
    split2(hi.z, hi.w, q0.w, q1.w, q2.w);
}

// ---- "f16x3" (NPROD == 2): two fp16 pieces per fp32 value, three products on v_mfma_f32_32x32x16_f16 -------------------------------
// An fp32 value scaled by a power of two splits into two fp16 pieces, a s = h + l + O(2^-22 |a s|) (h = round-to-nearest fp16, 11-bit
// significand; l = fp16 of the exact residual), and h h' + (h l' + l h') reproduces the product to 2^-22 -- 64 x tighter than the three
// bf16 products of "bf16x3" (2^-16) at the SAME three MFMAs per k-step and the same two-piece data path.  What fp16 lacks is range
// (2^-24 .. 65 504), so every operand TENSOR carries a power-of-two scale taken from its own absolute maximum (`k_amax`, an exact integer
// max over the fp32 bit patterns): 2^14 <= amax s < 2^15.  Elements down to 2^-17 amax keep all 22 bits, smaller ones an absolute error of
// 2^-39 amax -- below the fp32 rounding of any sum they enter.  Scales are powers of two, so scaling and un-scaling are exact; the epilogue
// multiplies the fp32 accumulator by the two inverse scales.
// An "|max| word" is 64 words, ONE PER 128-BYTE CACHE LINE (8 KiB in all): producers fold their maxima into word (workgroup id + wave) & 63.
// L2 atomics serialise per cache line at ~10 ns each -- thousands of waves folding into one line cost 20 - 50 us per launch (measured: +9 ms per
// train step from the LayerNorm epilogues alone, and no better with 64 words packed into two lines); spread over 64 lines they run in parallel
// channels.  Readers take the max over the 64 words with one gather load and a wave reduction.
constexpr int AMAX_STRIDE = 32;        // words between the 64 slots
__device__ inline uint32_t amax_line(const uint32_t *__restrict__ line)
{
    uint32_t m = line[(threadIdx.x & 63) * AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    return m;
}
__device__ inline void amax_fold(uint32_t *__restrict__ line, uint32_t m)     // m: this lane's maximum; one guarded atomic per wave
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    uint32_t *w = line + ((blockIdx.x + 7u * blockIdx.y + (threadIdx.x >> 6)) & 63u) * AMAX_STRIDE;
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, m);
}
__device__ inline float f16_scale(uint32_t amax_bits)
{
    const int e = (int)((amax_bits >> 23) & 0xff);
    if (e == 0 || e == 255) return 1.f;        // all-zero / denormal tensor; Inf / NaN inside (those propagate on their own)
    const int se = min(max(127 + 14 - (e - 127), 27), 227);       // s in [2^-100, 2^100]
    return __builtin_bit_cast(float, (uint32_t)se << 23);
}

// two (scaled) fp32 values -> their two fp16 pieces, each packed (lo = first value)
__device__ inline void split2h(float a, float b, uint32_t &p0, uint32_t &p1)
{
    // h = RNE fp16 of the pair; l = fp16 of the exact residuals a - h (v_fma_mix_f32 reads the fp16 halves in place: no v_cvt_f32_f16)
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(p0));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
}

// NPROD == 6 / 3: three bf16 pieces (the third unused by 3); NPROD == 2: two fp16 pieces of s * value (q2 is left alone)
template <int NPROD> __device__ inline void split8s(const float4 &lo, const float4 &hi, float s, uint4 &q0, uint4 &q1, uint4 &q2)
{
#ifdef VIT_EXP_NOSPLIT   /* experiment builds only (tools/exp_nosplit.sh): the operand as if it arrived already split -- WRONG results, the same loads / LDS traffic / MFMAs */
    if (NPROD == 2) { q0 = __builtin_bit_cast(uint4, lo); q1 = __builtin_bit_cast(uint4, hi); return; }
#endif
    if (NPROD == 2) {
        split2h(lo.x * s, lo.y * s, q0.x, q1.x);
        split2h(lo.z * s, lo.w * s, q0.y, q1.y);
        split2h(hi.x * s, hi.y * s, q0.z, q1.z);
        split2h(hi.z * s, hi.w * s, q0.w, q1.w);
    } else {
        split8(lo, hi, q0, q1, q2);
    }
}

template <int NPROD> __device__ inline f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c)
{
    if (NPROD == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// absolute maximum of a tensor as the bit pattern of |x| (monotone for non-negative floats; a NaN wins, so it stays visible): one
// guarded atomicMax per wave into a pre-zeroed 64-word line (amax_fold)
__global__ void __launch_bounds__(256) k_amax(const float *__restrict__ x, int64_t n, uint32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t m = 0;
    auto fold = [&](const uint4 &v) { m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu)); };
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
        const int64_t n4 = n >> 2;
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
            const uint4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
            fold(v0); fold(v1); fold(v2); fold(v3);
        }
        for (; i < n4; i += stride) fold(x4[i]);
        for (int64_t j = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = max(m, __builtin_bit_cast(uint32_t, x[j]) & 0x7fffffffu);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = max(m, __builtin_bit_cast(uint32_t, x[i]) & 0x7fffffffu);
    }
    amax_fold(out, m);
}

// SPLITK: gridDim.y workgroups share one output tile, each contracting its own range of K slabs and adding its partial
// sums into a pre-zeroed `out` with fp32 atomics (bias / residual enter through split 0; no activation).  Used when the
// tile count alone cannot fill the chip: the 257..514-row GEMMs of batch-1 inference give 48-160 tiles for 256 CUs.
template <int ACT, int TN, bool SPLITK, int NPROD>
__global__ void __launch_bounds__(256, 3) k_linear_x6(const float *__restrict__ x, const uint4 *__restrict__ wp,
                                                      const float *__restrict__ bias, const float *__restrict__ residual,
                                                      float *__restrict__ out, float *__restrict__ pre, int M, int N, int K,
                                                      const uint32_t *__restrict__ amax_x, const uint32_t *__restrict__ amax_w,
                                                      uint32_t *__restrict__ amax_out)
{
    constexpr int BN = 64 * TN;
    uint32_t omax = 0;          // |max| of the values this lane stores (published to *amax_out: the consumer's f16x3 scale without a pass of its own)
    // f16x3: scale of the activation tensor (applied while it is split) and the two inverse scales of the epilogue
    float sx = 1.f, ix = 1.f, iw = 1.f;
    if (NPROD == 2) { sx = f16_scale(amax_line(amax_x)); ix = 1.f / sx; iw = 1.f / f16_scale(amax_line(amax_w)); }
    // (the 64-row B tile is allocated at the 128-row size: keeps the narrow variant at three workgroups per CU; four
    // thrash the L2 on the N = 1024 layers: measured -5 %)
    __shared__ uint4 sA[2][BM * ROWQ], sB[2][128 * ROWQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // loader mapping: thread -> (tile row, 8-wide k group)
    const int lrow = tid >> 1, kg = tid & 1;
    const int KG = K >> 3;                                             // k groups per weight row
    // rows past M / N are CLAMPED, not predicated: their products land in accumulator rows / columns the epilogue never
    // stores, and branch-free loads let the compiler count outstanding loads exactly (s_waitcnt vmcnt(5) instead of 0:
    // the younger stage stays in flight across the LDS store of the older one)
    const float *xa = x + (int64_t)min(m0 + lrow, M - 1) * K + kg * 8;
    // B loader: 128-row tile -> thread = (row, k group), 3 pieces each.  64-row tile (TN = 1): thread = (row, k group,
    // part): part 0 moves pieces 0 and 1, part 1 moves piece 2 (twice, same address): every thread issues the same two
    // loads and two LDS stores, no predication
    const int brow = TN == 2 ? lrow : (tid >> 2), bkg = TN == 2 ? kg : ((tid >> 1) & 1);
    // (three-product mode never reads piece 2: the part-1 threads then duplicate their partner's pieces 0 / 1 -- same cache lines, same
    // LDS words -- instead of fetching it)
    const int pc0 = TN == 2 ? 0 : (((tid & 1) && NPROD == 6) ? 2 : 0), pc1 = TN == 2 ? 1 : (((tid & 1) && NPROD == 6) ? 2 : 1);
    const uint4 *wb = wp + ((int64_t)min(n0 + brow, N - 1) * KG + bkg) * 3;
    // two register stages: the global loads of slab k+2 are in flight while slab k feeds the MFMAs (one slab of MFMA work,
    // ~0.35 us, is shorter than the L2/HBM latency, so a single stage leaves the wave waiting at every LDS store)
#define X6_SPLIT8(lo, hi, q0, q1, q2) split8s<NPROD>(lo, hi, sx, q0, q1, q2)
    struct Stage { float4 a0, a1; uint4 b0, b1, b2; };
    Stage st0, st1;
    st0.b0 = st0.b1 = st0.b2 = st1.b0 = st1.b1 = st1.b2 = make_uint4(0, 0, 0, 0);
#define X6_GLOAD(S, k0)                                                                                              \
    do {                                                                                                             \
        const int k_ = min((k0), K - BK);   /* past the end: re-load the last slab (never consumed) */                \
        S.a0 = *reinterpret_cast<const float4 *>(xa + k_); S.a1 = *reinterpret_cast<const float4 *>(xa + k_ + 4);     \
        const uint4 *p_ = wb + (k_ >> 3) * 3;                                                                        \
        S.b0 = p_[pc0]; S.b1 = p_[pc1];                                                                              \
        if (TN == 2 && NPROD == 6) S.b2 = p_[2];                                                                     \
    } while (0)
#define X6_LSTORE(buf, S)                                                                                            \
    do {                                                                                                             \
        uint4 q0_, q1_, q2_;                                                                                         \
        X6_SPLIT8(S.a0, S.a1, q0_, q1_, q2_);                                                                        \
        uint4 *pa_ = sA[buf] + lrow * ROWQ;                                                                          \
        pa_[swz(lrow, kg * 3 + 0)] = q0_; pa_[swz(lrow, kg * 3 + 1)] = q1_; if (NPROD == 6) pa_[swz(lrow, kg * 3 + 2)] = q2_;   /* three-product mode never reads the third piece */        \
        uint4 *pb_ = sB[buf] + brow * ROWQ;                                                                          \
        pb_[swz(brow, bkg * 3 + pc0)] = S.b0; pb_[swz(brow, bkg * 3 + pc1)] = S.b1;                                  \
        if (TN == 2 && NPROD == 6) pb_[swz(brow, bkg * 3 + 2)] = S.b2;                                               \
    } while (0)

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};

    auto compute = [&](int buf) {
        const uint4 *a = sA[buf] + (wm * 64 + col) * ROWQ;          // rows wm*64 + 32 i + col, k group = half
        const uint4 *b = sB[buf] + (wn * 32 * TN + col) * ROWQ;
        bf16x8 fa[2][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, a[i * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = __builtin_bit_cast(bf16x8, b[j * 32 * ROWQ + swz(col, half * 3 + p)]);
        // smallest partial products first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                if (NPROD == 6) {   // the three 2^-16-level products; NPROD == 3 ("bf16x3") leaves them out
                    c = mma<NPROD>(fa[i][2], fb[j][0], c);
                    c = mma<NPROD>(fa[i][1], fb[j][1], c);
                    c = mma<NPROD>(fa[i][0], fb[j][2], c);
                }
                c = mma<NPROD>(fa[i][1], fb[j][0], c);
                c = mma<NPROD>(fa[i][0], fb[j][1], c);
                c = mma<NPROD>(fa[i][0], fb[j][0], c);
                acc[i][j] = c;
            }
    };

    int nk = K / BK;
    if (SPLITK) {   // this workgroup's slab range [k_lo, k_lo + nk)
        const int S = gridDim.y, s_ = blockIdx.y, base = nk / S, rem = nk % S;
        const int k_lo = s_ * base + min(s_, rem);
        nk = base + (s_ < rem ? 1 : 0);
        xa += (int64_t)k_lo * BK;
        wb += (int64_t)k_lo * (BK >> 3) * 3;
        K -= k_lo * BK;                  // the clamp in X6_GLOAD is relative to the shifted pointers
    }
    X6_GLOAD(st0, 0);
    X6_GLOAD(st1, BK);
    X6_LSTORE(0, st0);
    X6_GLOAD(st0, 2 * BK);
    __syncthreads();
    // iteration kt: MFMAs on LDS buffer kt&1; stage (kt+1)&1 holds slab kt+1 -> LDS; its registers then take slab kt+3
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        compute(0);
        X6_LSTORE(1, st1);
        X6_GLOAD(st1, (kt + 3) * BK);
        __syncthreads();
        compute(1);
        X6_LSTORE(0, st0);       // (the store after the last slab writes a buffer nobody reads)
        X6_GLOAD(st0, (kt + 4) * BK);
        __syncthreads();
    }
    if (kt < nk) compute(0);   // odd number of slabs
#undef X6_GLOAD
#undef X6_LSTORE

    // acc[i][j]: lane column n = n0 + wn*32*TN + 32 j + col ; register r = row m0 + wm*64 + 32 i + (r&3) + 8 (r>>2) + 4 half
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (32 * TN) + 32 * j + col;
        if (n >= N) continue;
        const bool first = !SPLITK || blockIdx.y == 0;
        const float bv = (bias && first) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const int64_t o = (int64_t)m * N + n;
                float t = acc[i][j][r];
                if (NPROD == 2) t = t * ix * iw;
                t += bv;
                if (SPLITK) {
                    if (residual && first) t += residual[o];
                    atomicAdd(out + o, t);
                    continue;
                }
                if (ACT == 2) {          // input-gradient GEMM of the layer BEHIND a GELU: out = (dY . W) * gelu'(pre), `residual` carries pre
                    t *= gelu_grad_exact(residual[o]);
                    out[o] = t;
                    omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu);
                    continue;
                }
                if (pre) pre[o] = t;
                if (ACT == 1) t = gelu_exact(t);
                if (residual) t += residual[o];
                out[o] = t;
                omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu);
            }
        }
    }
    if (!SPLITK && amax_out) {   // (wave-uniform branch; one atomic per wave)
        amax_fold(amax_out, omax);
    }
}


// Weight gradient dW (N,K) = dY^T (N,M) . X (M,K) [+ bias gradient db (N) = column sums of dY] on the same bf16x6 core.
// Both operands are activations whose contraction index m is the SLOW dimension in memory, so the loader reads them as
// coalesced rows (lane = output row n / column k, eight 4-byte loads walk m) and the transposition happens for free in
// the register -> LDS step: a thread's eight m-values of one column are exactly one 8-wide k group of the operand.  The
// output is small (N x K) and the contraction long (M = 4 000..5 000), so gridDim.y workgroups split M and add their
// partial tiles with fp32 atomics into the zeroed dW; workgroups of the first K-tile also accumulate db.
template <int NPROD>
__global__ void __launch_bounds__(256, 2) k_wgrad_x6(const float *__restrict__ dy, const float *__restrict__ x,
                                                     float *__restrict__ dw, float *__restrict__ dbias, int M, int N, int K,
                                                     int accumulate, const uint32_t *__restrict__ amax_dy, const uint32_t *__restrict__ amax_x)
{
    constexpr int TN = 2, BN = 128;
    float sa = 1.f, sb = 1.f, ia = 1.f, ib = 1.f;          // f16x3: tensor scales of dY and X, their inverses for the epilogue
    if (NPROD == 2) { sa = f16_scale(amax_line(amax_dy)); sb = f16_scale(amax_line(amax_x)); ia = 1.f / sa; ib = 1.f / sb; }
    __shared__ uint4 sA[2][BM * ROWQ], sB[2][BN * ROWQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = (K + BN - 1) / BN, tiles_n = (N + BM - 1) / BM;
    int tn_, tk_;
    tile_of_block(blockIdx.x, tiles_n, tiles_k, tn_, tk_);
    const int n0 = tn_ * BM, k0 = tk_ * BN;

    // slab range of this split
    const int nslab = (M + BK - 1) / BK, S = gridDim.y, sp = blockIdx.y;
    const int base = nslab / S, rem = nslab % S;
    const int s_lo = sp * base + min(sp, rem), nk = base + (sp < rem ? 1 : 0);

    // column of the tile, 8-row group of the slab (the group is wave-uniform: waves 0-1 / 2-3)
    const int c = tid & 127, g = __builtin_amdgcn_readfirstlane(tid >> 7);
    // Buffer loads: the row walk (m * N * 4 bytes) is a wave-uniform scalar added to the thread's constant column offset -- one 32-bit
    // add per load instead of 64-bit address arithmetic (r03: the flat-pointer form spent 2/3 of its VALU cycles on v_mad_i64 /
    // v_lshl_add_u64, twice the MFMA time in three-product mode).  The descriptors start at the split's first row and end at row M:
    // rows past M read as zero in hardware, so there is no clamp and no mask.
    const int64_t row0 = (int64_t)s_lo * BK;
    const int64_t left = max((int64_t)M - row0, (int64_t)0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dy + row0 * N), 0, (int)min(left * N * 4, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + row0 * K), 0, (int)min(left * K * 4, (int64_t)0x7fffffff), 0x00020000);
    const int voa = min(n0 + c, N - 1) * 4, vob = min(k0 + c, K - 1) * 4;
    const int N4 = N * 4, K4 = K * 4;
    const bool want_bias = dbias != nullptr && tk_ == 0;          // (workgroup-uniform)
    struct Stage { float a[8], b[8]; };
    Stage st0, st1;
    float bsum = 0.f;
#define W6_GLOAD(S_, slab)                                                                                            \
    do {                                                                                                              \
        const int m_ = (slab) * BK + g * 8;            /* row within the split (past its end: loaded, never consumed) */ \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                               \
            S_.a[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, voa + (m_ + r) * N4, 0, 0));   \
            S_.b[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, vob + (m_ + r) * K4, 0, 0));   \
        }                                                                                                             \
    } while (0)
#define W6_LSTORE(buf, S_, slab)                                                                                      \
    do {                                                                                                              \
        {   /* (branch-free: one scheduling region from barrier to barrier) */                                        \
            const float keep_ = (want_bias && (slab) < nk) ? 1.f : 0.f;                                               \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) bsum = fmaf(keep_, S_.a[r], bsum);                           \
        }                                                                                                             \
        uint4 q0_, q1_, q2_;                                                                                          \
        split8s<NPROD>(make_float4(S_.a[0], S_.a[1], S_.a[2], S_.a[3]), make_float4(S_.a[4], S_.a[5], S_.a[6], S_.a[7]), sa, q0_, q1_, q2_); \
        uint4 *p_ = sA[buf] + c * ROWQ;                                                                               \
        p_[swz_t(c, g * 3 + 0)] = q0_; p_[swz_t(c, g * 3 + 1)] = q1_; if (NPROD == 6) p_[swz_t(c, g * 3 + 2)] = q2_;                        \
        split8s<NPROD>(make_float4(S_.b[0], S_.b[1], S_.b[2], S_.b[3]), make_float4(S_.b[4], S_.b[5], S_.b[6], S_.b[7]), sb, q0_, q1_, q2_); \
        p_ = sB[buf] + c * ROWQ;                                                                                      \
        p_[swz_t(c, g * 3 + 0)] = q0_; p_[swz_t(c, g * 3 + 1)] = q1_; if (NPROD == 6) p_[swz_t(c, g * 3 + 2)] = q2_;                        \
    } while (0)

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};
    auto compute = [&](int buf) {
        const uint4 *a = sA[buf] + (wm * 64 + col) * ROWQ;
        const uint4 *b = sB[buf] + (wn * 64 + col) * ROWQ;
        bf16x8 fa[2][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, a[i * 32 * ROWQ + swz_t(col, half * 3 + p)]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = __builtin_bit_cast(bf16x8, b[j * 32 * ROWQ + swz_t(col, half * 3 + p)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 cc = acc[i][j];
                if (NPROD == 6) {   // the three 2^-16-level products; NPROD == 3 ("bf16x3") leaves them out
                    cc = mma<NPROD>(fa[i][2], fb[j][0], cc);
                    cc = mma<NPROD>(fa[i][1], fb[j][1], cc);
                    cc = mma<NPROD>(fa[i][0], fb[j][2], cc);
                }
                cc = mma<NPROD>(fa[i][1], fb[j][0], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][1], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][0], cc);
                acc[i][j] = cc;
            }
    };

    // (r03, measured and dropped: four register stages instead of two -- one workgroup alone needs 0.65 us per 16-row slab either way,
    // so that chain is not memory latency; the extra 32 VGPRs only cost the third resident workgroup per CU)
    if (nk > 0) {
        W6_GLOAD(st0, 0);
        W6_GLOAD(st1, 1);
        W6_LSTORE(0, st0, 0);
        W6_GLOAD(st0, 2);
        __syncthreads();
        int kt = 0;
        // Interleave request to the scheduler: the slab's MFMAs are independent of the next slab's split (VALU), its LDS stores and the
        // loads of the slab after; left alone the compiler emits them phase by phase and every pipe idles while another works
        // (r03 PMC, three-product mode: MFMA 34 %, LDS 34 %, VALU 25 %, TA 31 % busy -- they add up to one)
#define W6_SCHED()                                                                                                    \
    do {                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, NPROD == 6 ? 12 : 8, 0);                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < (NPROD == 6 ? 24 : 12); ++q_) {                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
            __builtin_amdgcn_sched_group_barrier(0x002, NPROD == 6 ? 5 : 6, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x020, NPROD == 6 ? 1 : 2, 0);                                       \
        }                                                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x200, NPROD == 6 ? 6 : 4, 0);                                           \
    } while (0)
        for (; kt + 1 < nk; kt += 2) {
            compute(0);
            W6_LSTORE(1, st1, kt + 1);
            W6_GLOAD(st1, kt + 3);
            W6_SCHED();
            __syncthreads();
            compute(1);
            W6_LSTORE(0, st0, kt + 2);   // (past the split's last slab: staged, never multiplied)
            W6_GLOAD(st0, kt + 4);
            W6_SCHED();
            __syncthreads();
        }
#undef W6_SCHED
        if (kt < nk) compute(0);
    }
#undef W6_GLOAD
#undef W6_LSTORE

    // acc[i][j]: lane column = k0 + wn*64 + 32 j + col ; register r = row n0 + wm*64 + 32 i + (r&3) + 8 (r>>2) + 4 half
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int k = k0 + wn * 64 + 32 * j + col;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n >= N) continue;
                float *o = dw + (int64_t)n * K + k;
                const float t = NPROD == 2 ? acc[i][j][r] * ia * ib : acc[i][j][r];
                if (!single) atomicAdd(o, t);
                else if (accumulate) *o += t;     // one workgroup owns the tile: plain read-add-write
                else *o = t;
            }
        }
    }
    if (want_bias && n0 + c < N) atomicAdd(dbias + n0 + c, bsum);
}

// (r03, measured and removed: the same weight gradient with PRODUCER and CONSUMER waves -- 8 waves per workgroup, waves 4-7 run the
// transposing loader one step ahead, waves 0-3 read fragments of slab j-1 while multiplying slab j-2, one bare s_barrier per slab, three
// LDS images.  Ablation had shown the two chains of this kernel do not overlap: loader alone 112 us, LDS reads + MFMAs alone 109 us,
// together 180 us on the 4096 x 1024 output.  Bit-exact, but one such workgroup per CU needs 0.54 us per slab against 0.63 us for one
// workgroup of this kernel and 0.51 us per slab-per-CU for this kernel at three workgroups per CU: 195 vs 165 us.  Each chain on its own
// is latency-bound at one wave per SIMD (starting co-resident workgroups of THIS kernel half a slab apart changes nothing either), and hipcc drains the producer's register prefetch at the loop edge (v_mov copies of in-flight
// loads behind s_waitcnt vmcnt(1..9)) in every formulation tried: asm / library barrier, sched_barrier pins, two or three stages.)
// 3x3 (pad 1, stride 1) and 1x1 convolution of the DPT heads (E9) as an implicit GEMM on the same bf16x6 core:
//     out[b][co][y][x] = bias[co] + sum_{tap, ci} w[co][tap][ci] * f(in[b][ci][y + ky - 1][x + kx - 1]),   f = id or ReLU
// GEMM view: A rows = output channels (the pre-split weight, k = tap * Ci + ci, loaded like the Linear's weight),
// B rows = output pixels, K = taps * Ci.  NCHW activations need no im2col and no layout change: the B loader is the
// transposing loader of the weight-gradient kernel (lane = pixel, eight 4-byte loads walk the channels of one tap, lanes
// are consecutive pixels = consecutive addresses), borders are clamped addresses + a zero mask.  D[co][pixel] has the
// pixel on the lane, so every accumulator register stores a coalesced run of one output channel.
// Epilogue: + bias, + residual (the skip connection of a ResidualConvUnit), or -- `gate` -- the ReLU mask of the
// backward pass; RELU_IN applies the unit's ReLU while the activations are staged (no separate ReLU pass / tensor).
template <int KS, bool RELU_IN, int NPROD>
__global__ void __launch_bounds__(256, 2) k_conv_x6(const float *__restrict__ in, const uint4 *__restrict__ wp,
                                                    const float *__restrict__ bias, const float *__restrict__ residual,
                                                    float *__restrict__ out, int B, int Ci, int Co, int H, int W, int gate,
                                                    const uint32_t *__restrict__ amax_w, const uint32_t *__restrict__ amax_in)
{
    constexpr int TN = 2, BN = 128, TAPS = KS * KS;
    float sb = 1.f, ia = 1.f, ib = 1.f;                     // f16x3: activation scale (rides in the border mask), inverse scales
    if (NPROD == 2) { sb = f16_scale(amax_line(amax_in)); ib = 1.f / sb; ia = 1.f / f16_scale(amax_line(amax_w)); }
    __shared__ uint4 sA[2][BM * ROWQ], sB[2][BN * ROWQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int HW = H * W;
    const int64_t NP = (int64_t)B * HW;
    const int tiles_m = (Co + BM - 1) / BM, tiles_n = (int)((NP + BN - 1) / BN);
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int co0 = tm * BM;
    const int64_t p0 = (int64_t)tn * BN;
    const int K = TAPS * Ci, KG = K >> 3;
    // gridDim.y > 1: the K = taps * Ci slabs are split across workgroups that add their partial tiles into the zeroed
    // output with fp32 atomics (small images: too few output tiles to fill the chip, and 144 slabs in a row are a
    // 0.12 ms floor); bias / residual enter through split 0
    const int nslab = K / BK, S_ = gridDim.y, sp_ = blockIdx.y;
    const int s_lo = sp_ * (nslab / S_) + min(sp_, nslab % S_), nk = nslab / S_ + (sp_ < nslab % S_ ? 1 : 0);

    // A loader (weights): thread -> (row, k group), three 16-byte pieces
    const int lrow = tid >> 1, kg = tid & 1;
    const uint4 *wa = wp + ((int64_t)min(co0 + lrow, Co - 1) * KG + kg) * 3;
    // B loader (activations): thread -> (pixel of the tile, 8-channel group)
    const int c = tid & 127, g = tid >> 7;
    const int64_t pc = min(p0 + c, NP - 1);
    const int pb = (int)(pc / HW), pr = (int)(pc - (int64_t)pb * HW), py = pr / W, px = pr - py * W;
    const float *inb = in + (int64_t)pb * Ci * HW;

    // two register stages, written out as scalars (a struct with mixed members ended up in scratch memory)
    uint4 a0_0, a1_0, a2_0, a0_1, a1_1, a2_1;
    float b0_0, b1_0, b2_0, b3_0, b4_0, b5_0, b6_0, b7_0, b0_1, b1_1, b2_1, b3_1, b4_1, b5_1, b6_1, b7_1, m_0, m_1;
    const int64_t HW64 = HW;
    const int spt = Ci / BK;                                           // slabs per tap
    const float *in_g = inb + (int64_t)(g * 8) * HW;
#define C6_GLOAD(T, slab)                                                                                             \
    do {                                                                                                              \
        const int sl_ = s_lo + min((slab), nk - 1);                                                                   \
        const uint4 *q_ = wa + sl_ * 6; a0_##T = q_[0]; a1_##T = q_[1]; a2_##T = q_[2];                               \
        const int tap = sl_ / spt, ci = (sl_ - tap * spt) * BK;                                                       \
        const int ky_ = KS == 3 ? (tap * 11) >> 5 : 0, kx_ = KS == 3 ? tap - 3 * ky_ : 0;                             \
        const int ys = py + (KS == 3 ? ky_ - 1 : 0), xs = px + (KS == 3 ? kx_ - 1 : 0);                               \
        m_##T = (ys >= 0 && ys < H && xs >= 0 && xs < W) ? sb : 0.f;                                                  \
        const float *s_ = in_g + (int64_t)ci * HW64 + (min(max(ys, 0), H - 1) * W + min(max(xs, 0), W - 1));         \
        b0_##T = s_[0]; b1_##T = s_[HW64]; b2_##T = s_[2 * HW64]; b3_##T = s_[3 * HW64];                              \
        b4_##T = s_[4 * HW64]; b5_##T = s_[5 * HW64]; b6_##T = s_[6 * HW64]; b7_##T = s_[7 * HW64];                   \
    } while (0)
#define C6_LSTORE(buf, T)                                                                                             \
    do {                                                                                                              \
        uint4 *pa_ = sA[buf] + lrow * ROWQ;                                                                           \
        pa_[swz(lrow, kg * 3 + 0)] = a0_##T; pa_[swz(lrow, kg * 3 + 1)] = a1_##T; if (NPROD == 6) pa_[swz(lrow, kg * 3 + 2)] = a2_##T; \
        const float lo_ = RELU_IN ? 0.f : -3.0e38f, mm_ = m_##T;   /* border taps contribute zero (clamped loads) */    \
        uint4 q0_, q1_, q2_;                                                                                          \
        split8s<NPROD>(make_float4(fmaxf(b0_##T, lo_) * mm_, fmaxf(b1_##T, lo_) * mm_, fmaxf(b2_##T, lo_) * mm_, fmaxf(b3_##T, lo_) * mm_), \
               make_float4(fmaxf(b4_##T, lo_) * mm_, fmaxf(b5_##T, lo_) * mm_, fmaxf(b6_##T, lo_) * mm_, fmaxf(b7_##T, lo_) * mm_), \
               1.f, q0_, q1_, q2_);                                                                                   \
        uint4 *pb_ = sB[buf] + c * ROWQ;                                                                              \
        pb_[swz(c, g * 3 + 0)] = q0_; pb_[swz(c, g * 3 + 1)] = q1_; if (NPROD == 6) pb_[swz(c, g * 3 + 2)] = q2_;                     \
    } while (0)

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};
    auto compute = [&](int buf) {
        const uint4 *a = sA[buf] + (wm * 64 + col) * ROWQ;
        const uint4 *b = sB[buf] + (wn * 64 + col) * ROWQ;
        bf16x8 fa[2][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, a[i * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = __builtin_bit_cast(bf16x8, b[j * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 cc = acc[i][j];
                if (NPROD == 6) {   // the three 2^-16-level products; NPROD == 3 ("bf16x3") leaves them out
                    cc = mma<NPROD>(fa[i][2], fb[j][0], cc);
                    cc = mma<NPROD>(fa[i][1], fb[j][1], cc);
                    cc = mma<NPROD>(fa[i][0], fb[j][2], cc);
                }
                cc = mma<NPROD>(fa[i][1], fb[j][0], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][1], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][0], cc);
                acc[i][j] = cc;
            }
    };

    C6_GLOAD(0, 0);
    C6_GLOAD(1, 1);
    C6_LSTORE(0, 0);
    C6_GLOAD(0, 2);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        compute(0);
        C6_LSTORE(1, 1);
        C6_GLOAD(1, kt + 3);
        __syncthreads();
        compute(1);
        C6_LSTORE(0, 0);
        C6_GLOAD(0, kt + 4);
        __syncthreads();
    }
    if (kt < nk) compute(0);
#undef C6_GLOAD
#undef C6_LSTORE

    // acc[i][j]: lane = pixel p0 + wn*64 + 32 j + col ; register r = channel co0 + wm*64 + 32 i + (r&3) + 8 (r>>2) + 4 half
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t p = p0 + wn * 64 + 32 * j + col;
        if (p >= NP) continue;
        const int b_ = (int)(p / HW);
        const int64_t obase = (int64_t)b_ * Co * HW + (p - (int64_t)b_ * HW);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co >= Co) continue;
                const int64_t o = obase + (int64_t)co * HW;
                const bool first = blockIdx.y == 0;
                float t = (NPROD == 2 ? acc[i][j][r] * ia * ib : acc[i][j][r]) + ((bias && first) ? bias[co] : 0.f);
                // gate: `residual` is the forward input of a ReLU-fused convolution and this launch computes its input
                // gradient: dX = (x > 0) ? conv^T(dY) : 0 (every K split gates its own partial sum)
                if (residual) { const float rv = residual[o]; if (gate) t = rv > 0.f ? t : 0.f; else if (first) t += rv; }
                if (gridDim.y == 1) out[o] = t; else atomicAdd(out + o, t);
            }
        }
    }
}

// 3x3 convolution, HALO form (r04).  k_conv_x6 treats the nine taps as nine independent K slabs: every 16-channel slab of every output
// pixel is fetched from global memory, ReLU'd, masked and SPLIT nine times (once per tap), and in the three-product modes that B-side
// data path -- not the matrix pipe -- is what bounds the kernel.  Here a workgroup owns a 4 x 32 patch of output pixels of ONE image
// (x 128 output channels) and stages, once per 16-channel slab, the (4+2) x (32+2) halo patch of the input: one fetch, one ReLU / border
// mask / split per input element, laid out pixel-major in LDS ([halo pixel][k group][piece][8 channels], the layout of k_conv_x6's B
// image), so the B fragment of tap (ky, kx) is the same image read at row offset ky * 34 + kx.  A 32-lane fragment column is one patch
// row = 32 consecutive halo rows (+ the tap offset), which keeps the XOR swizzle of `swz` conflict-free for every tap.  The weight
// tiles (A: 128 channels x 16 k per tap and slab) stream through two register stages and two LDS images as in k_conv_x6.
// Per slab and workgroup: 204 halo pixels x 16 channels fetched and split (k_conv_x6: 9 x 128 x 16), 9 x (2 x 2 x NPROD) MFMAs per wave.
// Epilogue and split-K (gridDim.y) semantics are k_conv_x6's; `amax_out` publishes the |max| of the stored values (unsplit launches).
template <bool RELU_IN, int NPROD>
__global__ void __launch_bounds__(256, 2) k_conv3h_x6(const float *__restrict__ in, const uint4 *__restrict__ wp,
                                                      const float *__restrict__ bias, const float *__restrict__ residual,
                                                      float *__restrict__ out, int B, int Ci, int Co, int H, int W, int gate,
                                                      const uint32_t *__restrict__ amax_w, const uint32_t *__restrict__ amax_in,
                                                      uint32_t *__restrict__ amax_out)
{
    constexpr int TY = 4, TX = 32, HY = TY + 2, HX = TX + 2, HP = HY * HX;      // 204 halo pixels
    __shared__ uint4 sA[2][BM * ROWQ], sP[2][HP * ROWQ];                         // 24 KB + 38.25 KB
    float sb = 1.f, ia = 1.f, ib = 1.f;                     // f16x3: activation scale (rides in the border mask), inverse scales
    if (NPROD == 2) { sb = f16_scale(amax_line(amax_in)); ib = 1.f / sb; ia = 1.f / f16_scale(amax_line(amax_w)); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int HW = H * W;
    const int tiles_x = (W + TX - 1) / TX, tiles_img = tiles_x * ((H + TY - 1) / TY);
    const int tiles_m = (Co + BM - 1) / BM, tiles_n = B * tiles_img;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int co0 = tm * BM;
    const int b_ = tn / tiles_img, tr = tn - b_ * tiles_img;
    const int ty0 = (tr / tiles_x) * TY, tx0 = (tr % tiles_x) * TX;
    const int KG = (9 * Ci) >> 3, CG = Ci >> 3;
    // gridDim.y > 1: the Ci / 16 channel slabs (nine taps each) are split across workgroups (fp32 atomics into the zeroed output)
    const int nslab = Ci / BK, S_ = gridDim.y, sp_ = blockIdx.y;
    const int s_lo = sp_ * (nslab / S_) + min(sp_, nslab % S_), ns = nslab / S_ + (sp_ < nslab % S_ ? 1 : 0);
    const int nit = ns * 9;

    // A loader (weights): thread -> (row, k group), three 16-byte pieces; k index of (tap, slab) = tap * Ci + slab * 16
    const int lrow = tid >> 1, kg = tid & 1;
    const uint4 *wa = wp + ((int64_t)min(co0 + lrow, Co - 1) * KG + kg) * 3;
    // halo loader: 2 k groups x 204 pixels = 408 items of 8 channels; thread -> items tid and tid + 256 (the latter for tid < 152);
    // consecutive lanes = consecutive pixels of a halo row = consecutive addresses of one channel plane
    const float *inb = in + (int64_t)b_ * Ci * HW;
    int h_off[2], h_row[2], h_slot[2];
    float h_m[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int item = min(tid + e * 256, 2 * HP - 1);
        const int kq = item / HP, hp = item - kq * HP, hy = hp / HX, hx = hp - hy * HX;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        h_m[e] = (y >= 0 && y < H && x >= 0 && x < W) ? sb : 0.f;              // border / outside-image pixels contribute zero
        h_off[e] = (kq * 8) * HW + min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
        h_row[e] = hp; h_slot[e] = kq * 3;
    }
    const bool second = tid < 2 * HP - 256;
    float hv0[8], hv1[8];
    auto halo_gload = [&](int s) {      // slab s of this split -> registers
        const float *p0_ = inb + (int64_t)(s_lo + s) * BK * HW;
#pragma unroll
        for (int c_ = 0; c_ < 8; ++c_) {
            hv0[c_] = p0_[h_off[0] + c_ * HW];
            hv1[c_] = p0_[h_off[1] + c_ * HW];                                   // (tid >= 152: a duplicate of the last item, stored to the same place)
        }
    };
    auto halo_store = [&](int buf) {
        const float lo_ = RELU_IN ? 0.f : -3.0e38f;
        uint4 q0_, q1_, q2_;
        {
            const float m_ = h_m[0];
            split8s<NPROD>(make_float4(fmaxf(hv0[0], lo_) * m_, fmaxf(hv0[1], lo_) * m_, fmaxf(hv0[2], lo_) * m_, fmaxf(hv0[3], lo_) * m_),
                           make_float4(fmaxf(hv0[4], lo_) * m_, fmaxf(hv0[5], lo_) * m_, fmaxf(hv0[6], lo_) * m_, fmaxf(hv0[7], lo_) * m_), 1.f, q0_, q1_, q2_);
            uint4 *pp_ = sP[buf] + h_row[0] * ROWQ;
            pp_[swz(h_row[0], h_slot[0] + 0)] = q0_; pp_[swz(h_row[0], h_slot[0] + 1)] = q1_; if (NPROD == 6) pp_[swz(h_row[0], h_slot[0] + 2)] = q2_;
        }
        if (second) {
            const float m_ = h_m[1];
            split8s<NPROD>(make_float4(fmaxf(hv1[0], lo_) * m_, fmaxf(hv1[1], lo_) * m_, fmaxf(hv1[2], lo_) * m_, fmaxf(hv1[3], lo_) * m_),
                           make_float4(fmaxf(hv1[4], lo_) * m_, fmaxf(hv1[5], lo_) * m_, fmaxf(hv1[6], lo_) * m_, fmaxf(hv1[7], lo_) * m_), 1.f, q0_, q1_, q2_);
            uint4 *pp_ = sP[buf] + h_row[1] * ROWQ;
            pp_[swz(h_row[1], h_slot[1] + 0)] = q0_; pp_[swz(h_row[1], h_slot[1] + 1)] = q1_; if (NPROD == 6) pp_[swz(h_row[1], h_slot[1] + 2)] = q2_;
        }
    };
    // two register stages of weight tiles (written out as scalars: see k_conv_x6)
    uint4 a0_0, a1_0, a2_0, a0_1, a1_1, a2_1;
    a2_0 = a2_1 = make_uint4(0, 0, 0, 0);
#define H6_AGLOAD(T, it_)                                                                                              \
    do {                                                                                                              \
        const int i_ = min((it_), nit - 1), s_ = i_ / 9, tp_ = i_ - s_ * 9;   /* past the end: the last tile again (never consumed) */ \
        const uint4 *q_ = wa + (int64_t)(tp_ * CG + (s_lo + s_) * 2) * 3;                                             \
        a0_##T = q_[0]; a1_##T = q_[1]; if (NPROD == 6) a2_##T = q_[2];                                               \
    } while (0)
#define H6_ASTORE(buf, T)                                                                                             \
    do {                                                                                                              \
        uint4 *pa_ = sA[buf] + lrow * ROWQ;                                                                           \
        pa_[swz(lrow, kg * 3 + 0)] = a0_##T; pa_[swz(lrow, kg * 3 + 1)] = a1_##T; if (NPROD == 6) pa_[swz(lrow, kg * 3 + 2)] = a2_##T; \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
    // halo row of this lane's pixel in fragment column j: patch row 2 wn + j, patch column `col`  (tap (ky, kx) adds ky * HX + kx)
    const int hp0 = (2 * wn) * HX + col;
    auto compute = [&](int abuf, int pbuf, int tap) {
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const uint4 *a = sA[abuf] + (wm * 64 + col) * ROWQ;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, a[i * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int hp = hp0 + (j + ky) * HX + kx;
            const uint4 *b = sP[pbuf] + hp * ROWQ;
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = __builtin_bit_cast(bf16x8, b[swz(hp, half * 3 + p)]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 cc = acc[i][j];
                if (NPROD == 6) {
                    cc = mma<NPROD>(fa[i][2], fb[j][0], cc);
                    cc = mma<NPROD>(fa[i][1], fb[j][1], cc);
                    cc = mma<NPROD>(fa[i][0], fb[j][2], cc);
                }
                cc = mma<NPROD>(fa[i][1], fb[j][0], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][1], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][0], cc);
                acc[i][j] = cc;
            }
    };

    if (nit > 0) {
        halo_gload(0);
        H6_AGLOAD(0, 0);
        H6_AGLOAD(1, 1);
        halo_store(0);
        H6_ASTORE(0, 0);
        H6_AGLOAD(0, 2);
        __syncthreads();
        // iteration `it` = (slab it / 9, tap it % 9): MFMAs on weight image it & 1 and patch image slab & 1; weight stage (it + 1) & 1
        // -> LDS, its registers then fetch tile it + 3; the NEXT slab's halo is fetched at tap 0 and split / stored at tap 4
        int s = 0, tap = 0;
#define H6_STEP(T_CUR, T_NXT, it_)                                                                                     \
    do {                                                                                                              \
        if (tap == 0 && s + 1 < ns) halo_gload(s + 1);                                                                \
        compute(T_CUR, s & 1, tap);                                                                                   \
        H6_ASTORE(T_NXT, T_NXT);                                                                                      \
        H6_AGLOAD(T_NXT, (it_) + 3);                                                                                  \
        if (tap == 4 && s + 1 < ns) halo_store((s + 1) & 1);                                                          \
        __syncthreads();                                                                                              \
        if (++tap == 9) { tap = 0; ++s; }                                                                             \
    } while (0)
        int it = 0;
        for (; it + 1 < nit; it += 2) {
            H6_STEP(0, 1, it);
            H6_STEP(1, 0, it + 1);
        }
        if (it < nit) compute(0, s & 1, tap);      // (nit odd: the last tile sits in weight image 0)
#undef H6_STEP
    }
#undef H6_AGLOAD
#undef H6_ASTORE

    // acc[i][j]: lane = pixel (ty0 + 2 wn + j, tx0 + col) ; register r = channel co0 + wm*64 + 32 i + (r&3) + 8 (r>>2) + 4 half
    uint32_t omax = 0;
    const bool first = blockIdx.y == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int y = ty0 + 2 * wn + j, x = tx0 + col;
        if (y >= H || x >= W) continue;
        const int64_t obase = (int64_t)b_ * Co * HW + (int64_t)y * W + x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co >= Co) continue;
                const int64_t o = obase + (int64_t)co * HW;
                float t = (NPROD == 2 ? acc[i][j][r] * ia * ib : acc[i][j][r]) + ((bias && first) ? bias[co] : 0.f);
                if (residual) { const float rv = residual[o]; if (gate) t = rv > 0.f ? t : 0.f; else if (first) t += rv; }
                if (gridDim.y == 1) { out[o] = t; omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu); }
                else atomicAdd(out + o, t);
            }
        }
    }
    if (amax_out && gridDim.y == 1) {
        amax_fold(amax_out, omax);
    }
}

// Weight (+ bias) gradient of that convolution:  dw[co][ci][tap] = sum_{b,y,x} dy[b][co][y][x] * f(in[b][ci][y+ky-1][x+kx-1]).
// GEMM view: A rows = co, B rows = r = tap * Ci + ci, contraction = the B*H*W output pixels, which are the CONTIGUOUS
// dimension of both operands (NCHW): both loaders are the Linear kernel's activation loader (thread = (row, 8-pixel
// group), 32 contiguous bytes), the tap shift is an offset on the B pointer plus a zero mask on the row / the first or
// last pixel of an image row.  The contraction is 10^5 pixels long and the output 256 x 2 304: gridDim.y workgroups
// split the pixels and accumulate with fp32 atomics into the zeroed dw (and db from the row sums of dy).
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16-byte load with 4-byte alignment (tap shift +-1)

template <int KS, bool RELU_IN, int NPROD>
__global__ void __launch_bounds__(256, 2) k_conv_wgrad_x6(const float *__restrict__ dy, const float *__restrict__ in,
                                                          float *__restrict__ dw, float *__restrict__ dbias, int B, int Ci,
                                                          int Co, int H, int W, const uint32_t *__restrict__ amax_dy,
                                                          const uint32_t *__restrict__ amax_in)
{
    constexpr int TN = 2, BN = 128, TAPS = KS * KS;
    float sa = 1.f, sb = 1.f, ia = 1.f, ib = 1.f;          // f16x3: tensor scales of dY and the input (the latter rides in the row mask)
    if (NPROD == 2) { sa = f16_scale(amax_line(amax_dy)); sb = f16_scale(amax_line(amax_in)); ia = 1.f / sa; ib = 1.f / sb; }
    __shared__ uint4 sA[2][BM * ROWQ], sB[2][BN * ROWQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int HW = H * W, R = TAPS * Ci;
    const int tiles_m = (Co + BM - 1) / BM, tiles_n = (R + BN - 1) / BN;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int co0 = tm * BM, r0 = tn * BN;

    const int spi = HW / BK;                                   // slabs per image
    const int nslab = B * spi, S = gridDim.y, sp = blockIdx.y;
    const int base = nslab / S, rem = nslab % S;
    const int s_lo = sp * base + min(sp, rem), nk = base + (sp < rem ? 1 : 0);

    const int lrow = tid >> 1, kg = tid & 1;
    // A: dy row co, 8 pixels;  B: in row (tap, ci) shifted by the tap, 8 pixels
    const int co = min(co0 + lrow, Co - 1);
    const int r = min(r0 + lrow, R - 1), tap = r / Ci, ci = r - tap * Ci;
    const int ky = KS == 3 ? tap / 3 - 1 : 0, kx = KS == 3 ? tap - (tap / 3) * 3 - 1 : 0;
    const int64_t total_in = (int64_t)B * Ci * HW;
    float4 a0_0, a1_0, a0_1, a1_1;
    f4u b0_0, b1_0, b0_1, b1_1;
    float my_0, my_1;      // row-validity mask of the stage (the 8 pixels of a group share one image row: W % 8 == 0)
    int x0_0, x0_1;
    float bsum = 0.f;
#define G6_GLOAD(T, slab)                                                                                             \
    do {                                                                                                              \
        const int sl_ = s_lo + min((slab), nk - 1);                                                                   \
        const int b_ = sl_ / spi, q_ = (sl_ - b_ * spi) * BK + kg * 8;          /* first pixel of the group */          \
        const float *pa_ = dy + ((int64_t)b_ * Co + co) * HW + q_;                                                    \
        a0_##T = *reinterpret_cast<const float4 *>(pa_); a1_##T = *reinterpret_cast<const float4 *>(pa_ + 4);         \
        const int y_ = q_ / W, xx_ = q_ - y_ * W;                                                                     \
        my_##T = (y_ + ky >= 0 && y_ + ky < H) ? sb : 0.f;                                                            \
        x0_##T = xx_ + kx;                                                                                            \
        int64_t o_ = ((int64_t)b_ * Ci + ci) * HW + q_ + ky * W + kx;                                                 \
        /* the +-1 tap shift can start one element before / end one element after the tensor: load the clamped run and  \
           slide it back (every other out-of-range run belongs to a masked row: only its address has to be legal) */    \
        const bool lo1_ = o_ == -1, hi1_ = o_ == total_in - 7;                                                        \
        o_ = min(max(o_, (int64_t)0), total_in - 8);                                                                  \
        const f4u u0_ = *reinterpret_cast<const f4u *>(in + o_), u1_ = *reinterpret_cast<const f4u *>(in + o_ + 4);   \
        b0_##T.x = lo1_ ? 0.f : (hi1_ ? u0_.y : u0_.x); b0_##T.y = lo1_ ? u0_.x : (hi1_ ? u0_.z : u0_.y);             \
        b0_##T.z = lo1_ ? u0_.y : (hi1_ ? u0_.w : u0_.z); b0_##T.w = lo1_ ? u0_.z : (hi1_ ? u1_.x : u0_.w);           \
        b1_##T.x = lo1_ ? u0_.w : (hi1_ ? u1_.y : u1_.x); b1_##T.y = lo1_ ? u1_.x : (hi1_ ? u1_.z : u1_.y);           \
        b1_##T.z = lo1_ ? u1_.y : (hi1_ ? u1_.w : u1_.z); b1_##T.w = lo1_ ? u1_.z : (hi1_ ? 0.f : u1_.w);             \
    } while (0)
#define G6_LSTORE(buf, T, slab)                                                                                       \
    do {                                                                                                              \
        const float live_ = (slab) < nk ? 1.f : 0.f;                                                                  \
        const float4 va0 = make_float4(a0_##T.x * live_, a0_##T.y * live_, a0_##T.z * live_, a0_##T.w * live_);       \
        const float4 va1 = make_float4(a1_##T.x * live_, a1_##T.y * live_, a1_##T.z * live_, a1_##T.w * live_);       \
        bsum += (va0.x + va0.y) + (va0.z + va0.w) + (va1.x + va1.y) + (va1.z + va1.w);                                 \
        uint4 q0_, q1_, q2_;                                                                                          \
        split8s<NPROD>(va0, va1, sa, q0_, q1_, q2_);                                                                  \
        uint4 *pa2_ = sA[buf] + lrow * ROWQ;                                                                          \
        pa2_[swz(lrow, kg * 3 + 0)] = q0_; pa2_[swz(lrow, kg * 3 + 1)] = q1_; if (NPROD == 6) pa2_[swz(lrow, kg * 3 + 2)] = q2_;      \
        const float lo_ = RELU_IN ? 0.f : -3.0e38f, mm_ = my_##T;                                                     \
        const float m0_ = (x0_##T >= 0) ? mm_ : 0.f, m7_ = (x0_##T + 7 < W) ? mm_ : 0.f;   /* only the ends can leave the row */ \
        split8s<NPROD>(make_float4(fmaxf(b0_##T.x, lo_) * m0_, fmaxf(b0_##T.y, lo_) * mm_, fmaxf(b0_##T.z, lo_) * mm_, fmaxf(b0_##T.w, lo_) * mm_), \
               make_float4(fmaxf(b1_##T.x, lo_) * mm_, fmaxf(b1_##T.y, lo_) * mm_, fmaxf(b1_##T.z, lo_) * mm_, fmaxf(b1_##T.w, lo_) * m7_), \
               1.f, q0_, q1_, q2_);                                                                                   \
        uint4 *pb2_ = sB[buf] + lrow * ROWQ;                                                                          \
        pb2_[swz(lrow, kg * 3 + 0)] = q0_; pb2_[swz(lrow, kg * 3 + 1)] = q1_; if (NPROD == 6) pb2_[swz(lrow, kg * 3 + 2)] = q2_;      \
    } while (0)

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};
    auto compute = [&](int buf) {
        const uint4 *a = sA[buf] + (wm * 64 + col) * ROWQ;
        const uint4 *b = sB[buf] + (wn * 64 + col) * ROWQ;
        bf16x8 fa[2][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = __builtin_bit_cast(bf16x8, a[i * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = __builtin_bit_cast(bf16x8, b[j * 32 * ROWQ + swz(col, half * 3 + p)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 cc = acc[i][j];
                if (NPROD == 6) {   // the three 2^-16-level products; NPROD == 3 ("bf16x3") leaves them out
                    cc = mma<NPROD>(fa[i][2], fb[j][0], cc);
                    cc = mma<NPROD>(fa[i][1], fb[j][1], cc);
                    cc = mma<NPROD>(fa[i][0], fb[j][2], cc);
                }
                cc = mma<NPROD>(fa[i][1], fb[j][0], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][1], cc);
                cc = mma<NPROD>(fa[i][0], fb[j][0], cc);
                acc[i][j] = cc;
            }
    };

    if (nk > 0) {
        G6_GLOAD(0, 0);
        G6_GLOAD(1, 1);
        G6_LSTORE(0, 0, 0);
        G6_GLOAD(0, 2);
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            compute(0);
            G6_LSTORE(1, 1, kt + 1);
            G6_GLOAD(1, kt + 3);
            __syncthreads();
            compute(1);
            G6_LSTORE(0, 0, kt + 2);
            G6_GLOAD(0, kt + 4);
            __syncthreads();
        }
        if (kt < nk) compute(0);
    }
#undef G6_GLOAD
#undef G6_LSTORE

    // acc[i][j]: lane = r0 + wn*64 + 32 j + col (tap, ci) ; register rr = co0 + wm*64 + 32 i + (rr&3) + 8 (rr>>2) + 4 half
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rr_ = r0 + wn * 64 + 32 * j + col;
        if (rr_ >= R) continue;
        const int tp = rr_ / Ci, c_ = rr_ - tp * Ci;
        float *o = dw + (int64_t)c_ * TAPS + tp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co_ = co0 + wm * 64 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * half;
                if (co_ >= Co) continue;
                atomicAdd(o + (int64_t)co_ * Ci * TAPS, NPROD == 2 ? acc[i][j][q] * ia * ib : acc[i][j][q]);
            }
        }
    }
    if (dbias && tn == 0 && co0 + lrow < Co) atomicAdd(dbias + co0 + lrow, bsum);
}

// w (R, C) row-major fp32 -> packed[r][c/8][piece][8] bf16.  transpose = 0: (r, c) = (row, col) of w, R x C = rows x cols.
// transpose = 1: packs w^T, i.e. output row r = column r of w, output k index = row of w (tiled through LDS so both the
// reads and the writes stay coalesced).
// NPROD == 2 ("f16x3"): two fp16 pieces of w * f16_scale(amax_line(amax)) in slots 0 / 1 (slot 2 is never read in that mode)
template <int NPROD>
__global__ void __launch_bounds__(256) k_split_rows(const float *__restrict__ w, uint4 *__restrict__ packed, int64_t groups,
                                                    const uint32_t *__restrict__ amax, uint32_t *__restrict__ tail)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-wide k group per thread
    if (NPROD == 2 && blockIdx.x == 0 && threadIdx.x < 64) tail[threadIdx.x * AMAX_STRIDE] = amax[threadIdx.x * AMAX_STRIDE];      // the line the kernels that read this image take their inverse scale from
    const float sw = NPROD == 2 ? f16_scale(amax_line(amax)) : 1.f;     // (whole waves: before any lane leaves)
    if (g >= groups) return;
    const float4 lo = reinterpret_cast<const float4 *>(w)[g * 2], hi = reinterpret_cast<const float4 *>(w)[g * 2 + 1];
    uint4 q0, q1, q2 = make_uint4(0, 0, 0, 0);
    split8s<NPROD>(lo, hi, sw, q0, q1, q2);
    packed[g * 3 + 0] = q0; packed[g * 3 + 1] = q1; if (NPROD != 2) packed[g * 3 + 2] = q2;
}

template <int NPROD>
__global__ void __launch_bounds__(256) k_split_transposed(const float *__restrict__ w, uint4 *__restrict__ packed, int rows,
                                                          int cols, const uint32_t *__restrict__ amax, uint32_t *__restrict__ tail)
{
    if (NPROD == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) tail[threadIdx.x * AMAX_STRIDE] = amax[threadIdx.x * AMAX_STRIDE];
    const float sw = NPROD == 2 ? f16_scale(amax_line(amax)) : 1.f;     // (whole waves: before any lane leaves)
    // tile: 64 source rows (-> k of the output) x 32 source columns (-> output rows)
    __shared__ float s[64][33];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        s[r][c] = (r0 + r < rows && c0 + c < cols) ? w[(int64_t)(r0 + r) * cols + c0 + c] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x >> 3, kg = threadIdx.x & 7;              // 32 output rows x 8 k groups
    if (c0 + c >= cols || r0 + kg * 8 >= rows) return;
    float4 lo = make_float4(s[kg * 8 + 0][c], s[kg * 8 + 1][c], s[kg * 8 + 2][c], s[kg * 8 + 3][c]);
    float4 hi = make_float4(s[kg * 8 + 4][c], s[kg * 8 + 5][c], s[kg * 8 + 6][c], s[kg * 8 + 7][c]);
    uint4 q0, q1, q2 = make_uint4(0, 0, 0, 0);
    split8s<NPROD>(lo, hi, sw, q0, q1, q2);
    uint4 *o = packed + ((int64_t)(c0 + c) * (rows >> 3) + (r0 >> 3) + kg) * 3;
    o[0] = q0; o[1] = q1; if (NPROD != 2) o[2] = q2;
}

// Both operand images of a convolution weight w (Co, Ci, k, k) straight from the parameter, in ONE launch (round 6; before: a permute + clone, a
// flip, a second permute + clone and two k_split_rows launches per weight and optimizer step -- ~290 framework copy launches of a C3 step):
//   forward image  rows = co, column = tap * Ci + ci               value w[co][ci][tap]                 (vit_conv_x6_fwd's weight)
//   dX image       rows = ci, column = tap * Co + co               value w[co][ci][k*k - 1 - tap]       (spatially flipped, channel-transposed)
// both in the row layout of vit_split_weight.  A workgroup owns 16 output channels x 16 input channels x all taps: each of the two images'
// 8-wide k groups (8 consecutive ci at one (co, tap) / 8 consecutive co at one (ci, tap)) lies inside it.  pdx may be null (forward image only).
constexpr int CW_T = 16, CW_TAPS_MAX = 9;
template <int NPROD>
__global__ void __launch_bounds__(256) k_split_conv_pair(const float *__restrict__ w, uint4 *__restrict__ pf, uint4 *__restrict__ pdx, int Co, int Ci, int kk,
                                                         const uint32_t *__restrict__ amax, uint32_t *__restrict__ tail_f, uint32_t *__restrict__ tail_dx)
{
    if (NPROD == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        const uint32_t a = amax[threadIdx.x * AMAX_STRIDE];
        tail_f[threadIdx.x * AMAX_STRIDE] = a;
        if (pdx) tail_dx[threadIdx.x * AMAX_STRIDE] = a;
    }
    const float sw = NPROD == 2 ? f16_scale(amax_line(amax)) : 1.f;
    __shared__ float s[CW_T][CW_T * CW_TAPS_MAX + 1];            // [co][ci * kk + tap]
    const int co0 = blockIdx.y * CW_T, ci0 = blockIdx.x * CW_T, run = CW_T * kk;
    for (int i = threadIdx.x; i < CW_T * run; i += 256) {         // per co: CW_T * kk contiguous floats
        const int co = i / run, e = i - co * run;
        const int ci = e / kk;
        s[co][e] = (co0 + co < Co && ci0 + ci < Ci) ? w[((int64_t)(co0 + co) * Ci + ci0) * kk + e] : 0.f;
    }
    __syncthreads();
    const int items = CW_T * kk * (CW_T / 8);
    {   // forward image
        const int KG = (kk * Ci) >> 3;
        for (int i = threadIdx.x; i < items; i += 256) {
            const int cg = i % (CW_T / 8), t = (i / (CW_T / 8)) % kk, co = i / ((CW_T / 8) * kk);
            if (co0 + co >= Co || ci0 + cg * 8 >= Ci) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s[co][(cg * 8 + j) * kk + t];
            uint4 q0, q1, q2 = make_uint4(0, 0, 0, 0);
            split8s<NPROD>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sw, q0, q1, q2);
            uint4 *o = pf + ((int64_t)(co0 + co) * KG + ((t * Ci + ci0) >> 3) + cg) * 3;
            o[0] = q0; o[1] = q1; if (NPROD != 2) o[2] = q2;
        }
    }
    if (pdx) {   // input-gradient image
        const int KG = (kk * Co) >> 3;
        for (int i = threadIdx.x; i < items; i += 256) {
            const int cg = i % (CW_T / 8), t = (i / (CW_T / 8)) % kk, ci = i / ((CW_T / 8) * kk);
            if (ci0 + ci >= Ci || co0 + cg * 8 >= Co) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s[cg * 8 + j][ci * kk + (kk - 1 - t)];
            uint4 q0, q1, q2 = make_uint4(0, 0, 0, 0);
            split8s<NPROD>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sw, q0, q1, q2);
            uint4 *o = pdx + ((int64_t)(ci0 + ci) * KG + ((t * Co + co0) >> 3) + cg) * 3;
            o[0] = q0; o[1] = q1; if (NPROD != 2) o[2] = q2;
        }
    }
}
}  // namespace x6

// Partial products per bf16x6 kernel launch: 6 (default, fp32 round-off accuracy) or 3 ("bf16x3": a0 b0 + a0 b1 + a1 b0, operands
// good to 2^-18 -- ~3.5e-6 of the output scale per GEMM, two orders tighter than the TF32 the reference enables, croco.py:13).
// Per HOST THREAD (thread_local) and read at launch time on the launching thread: a thread's set + launch pair cannot be
// interleaved with another thread's choice (a serving thread in bf16x3 beside a training thread in bf16x6).
// n == 2 selects "f16x3": TWO fp16 pieces per operand and the same three products on v_mfma_f32_32x32x16_f16 -- 2^-22 instead of 2^-16
// per product at the price of bf16x3; every operand tensor then needs its absolute maximum (vit_amax + vit_x6_set_operand_amax for
// activations; the weight's rides behind its packed image, see split_weight).
static thread_local int g_x6_products = 6;
static thread_local const uint32_t *g_amax_a = nullptr, *g_amax_b = nullptr;
int x6_set_products(int n)
{
    if (n != 2 && n != 3 && n != 6) return VIT_EINVAL;
    g_x6_products = n;
    return VIT_OK;
}
int x6_products() { return g_x6_products; }
// Device addresses of the |max| words (k_amax) of the ACTIVATION operands of the NEXT x6 launch on this host thread: a = x (Linear / conv
// forward and input-gradient launches), or a = dY, b = x (weight-gradient launches).  Consumed by that launch.
int x6_set_operand_amax(const void *a, const void *b)
{
    g_amax_a = static_cast<const uint32_t *>(a); g_amax_b = static_cast<const uint32_t *>(b);
    return VIT_OK;
}
static thread_local uint32_t *g_amax_out = nullptr;
// Device address of a ZEROED word that the NEXT vit_linear_x6_fwd / vit_linear_x6r_fwd launch on this thread fills with the |max| of the
// values it stores (its epilogue sees every one of them): the consumer of that output then needs no vit_amax pass.  Consumed by that
// launch (split-K launches run a vit_amax pass over their result instead).
int x6_set_output_amax(void *word) { g_amax_out = static_cast<uint32_t *>(word); return VIT_OK; }
uint32_t *x6_take_output_amax() { uint32_t *p = g_amax_out; g_amax_out = nullptr; return p; }
static void take_amax(const uint32_t *&a, const uint32_t *&b) { a = g_amax_a; b = g_amax_b; g_amax_a = g_amax_b = nullptr; }
void x6_take_amax(const uint32_t *&a, const uint32_t *&b) { take_amax(a, b); }      // (vit_gemm_x6r.hip)
// the weight's |max| word lives right behind its rows x cols x 6 bytes of pieces (vit_split_weight_bytes leaves room)
static const uint32_t *weight_amax(const void *packed, int rows, int cols)
{
    return reinterpret_cast<const uint32_t *>(static_cast<const char *>(packed) + (size_t)rows * (size_t)cols * 6);
}
static bool zero_fill(void *p, size_t bytes, hipStream_t stream);
int amax(const float *x, int64_t n, void *out, hipStream_t stream)     // *out must be zero before the launch
{
    if (!x || !out || n <= 0) return VIT_EINVAL;
    (void)hipGetLastError();
    const int64_t blocks = (n / 16 + 255) / 256;            // ~four 16-byte loads per lane; 2048 workgroups = 8 per CU at most
    hipLaunchKernelGGL(x6::k_amax, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, stream, x, n, static_cast<uint32_t *>(out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

// number of contraction splits for the weight-gradient kernels (2 resident workgroups per CU): one full round of 512
// workgroups when the tiles alone are fewer, otherwise whole multiples are left to the tile count; >= min_slabs per split
static int split_count(int tiles, int nslab, int min_slabs)
{
    // pick S in 1..128 that fills whole rounds of the 512 resident workgroups best; every extra split costs another pass
    // of atomics over the output tile (-1 % per split in the score)
    int best = 1;
    float best_score = -1.f;
    constexpr int slots = 512;
    for (int S = 1; S <= 128; ++S) {
        if (S > 1 && nslab / S < min_slabs) break;
        const int wg = tiles * S, rounds = (wg + slots - 1) / slots;
        const float score = (float)wg / (float)(rounds * slots) - 0.002f * (float)S;
        if (score > best_score) { best_score = score; best = S; }
    }
    return best;
}

// the two images of a convolution weight (Co, Ci, k, k), k in {1, 3}: see k_split_conv_pair.  packed_dx may be null.  Ci % 8 == 0; with a dX image
// also Co % 8 == 0.  f16x3: the weight's |max| word must be announced (vit_x6_set_operand_amax(word, NULL)).
int split_conv_weight_pair(const float *w, void *packed_fwd, void *packed_dx, int Co, int Ci, int ksize, hipStream_t stream)
{
    if (!w || !packed_fwd || Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3) || (Ci % 8) != 0 || (packed_dx && (Co % 8) != 0)) return VIT_EINVAL;
    (void)hipGetLastError();
    const int kk = ksize * ksize;
    const uint32_t *am, *unused;
    take_amax(am, unused);
    uint32_t *tf = const_cast<uint32_t *>(weight_amax(packed_fwd, Co, kk * Ci));
    uint32_t *td = packed_dx ? const_cast<uint32_t *>(weight_amax(packed_dx, Ci, kk * Co)) : nullptr;
    const dim3 grid((Ci + x6::CW_T - 1) / x6::CW_T, (Co + x6::CW_T - 1) / x6::CW_T);
    if (x6_products() == 2) {
        if (!am) return VIT_EINVAL;
        hipLaunchKernelGGL(x6::k_split_conv_pair<2>, grid, dim3(256), 0, stream, w, static_cast<uint4 *>(packed_fwd), static_cast<uint4 *>(packed_dx), Co, Ci, kk, am, tf, td);
    } else
        hipLaunchKernelGGL(x6::k_split_conv_pair<6>, grid, dim3(256), 0, stream, w, static_cast<uint4 *>(packed_fwd), static_cast<uint4 *>(packed_dx), Co, Ci, kk, am, tf, td);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int split_weight(const float *w, void *packed, int rows, int cols, int transpose, hipStream_t stream)
{
    if (!w || !packed || rows <= 0 || cols <= 0) return VIT_EINVAL;
    if ((transpose ? rows : cols) % 8 != 0) return VIT_EINVAL;
    (void)hipGetLastError();
    const bool f16 = x6_products() == 2;
    uint32_t *tail = const_cast<uint32_t *>(weight_amax(packed, rows, cols));
    // f16x3: the weight's own scale.  The caller may announce the weight's |max| word (vit_x6_set_operand_amax(word, NULL): one vit_amax pass
    // then serves both images of a weight, forward and transposed); otherwise it is computed here into the word behind the pieces
    const uint32_t *am, *unused;
    take_amax(am, unused);
    if (f16 && !am) {
        if (!zero_fill(tail, 8192, stream)) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
        const int rc = amax(w, (int64_t)rows * cols, tail, stream);
        if (rc != VIT_OK) return rc;
        am = tail;
    }
    if (!transpose) {
        const int64_t groups = (int64_t)rows * cols / 8;
        if (f16) hipLaunchKernelGGL(x6::k_split_rows<2>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), groups, am, tail);
        else hipLaunchKernelGGL(x6::k_split_rows<6>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), groups, am, tail);
    } else {
        if (f16) hipLaunchKernelGGL(x6::k_split_transposed<2>, dim3((cols + 31) / 32, (rows + 63) / 64), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), rows, cols, am, tail);
        else hipLaunchKernelGGL(x6::k_split_transposed<6>, dim3((cols + 31) / 32, (rows + 63) / 64), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), rows, cols, am, tail);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

static bool zero_fill(void *p, size_t bytes, hipStream_t stream)
{
    const size_t n = bytes / 4;
    if (n == 0) return true;
    const size_t groups = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(x6::k_zero_words, dim3((unsigned)(groups < 1 ? 1 : (groups > 2048 ? 2048 : groups))), dim3(256), 0, stream,
                       static_cast<uint32_t *>(p), n);
    return hipGetLastError() == hipSuccess;
}

int linear_x6_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N,
                  int K, int act, hipStream_t stream)
{
    if (!x || !wp || !out) return VIT_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % x6::BK) != 0 || act < 0 || act > 2 || (act == 2 && (!residual || pre))) return VIT_EINVAL;
    const int tm = (M + x6::BM - 1) / x6::BM;
    const bool narrow = tm * ((N + 127) / 128) < 640;
    const int tiles = tm * (narrow ? (N + 63) / 64 : (N + 127) / 128);
    const uint4 *w4 = static_cast<const uint4 *>(wp);
    const int np = x6_products();
    const uint32_t *am_x, *am_unused, *am_w = weight_amax(wp, N, K);
    take_amax(am_x, am_unused);
    uint32_t *am_out = x6_take_output_amax();
    if (np == 2 && !am_x) return VIT_EINVAL;      // f16x3 without the activation's |max|: refuse, never guess a scale
    (void)hipGetLastError();
    // split-K when the tiles cannot fill the 256 CUs x 3 resident workgroups: S = 2 / 4 / 8 ranges of >= 8 slabs each
    int S = 1;
    if (act == 0 && !pre && tiles <= 200) {   // (240 tiles unsplit beat 2 x 240 with the memset + atomics: measured)
        const int nk = K / x6::BK;
        while (S < 8 && tiles * S * 2 <= 768 && nk / (S * 2) >= 8) S *= 2;
    }
    if (S > 1) {
        if (!zero_fill(out, (size_t)M * N * sizeof(float), stream)) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
#define VIT_LAUNCH_X6S(TN, NP) hipLaunchKernelGGL((x6::k_linear_x6<0, TN, true, NP>), dim3(tiles, S), dim3(256), 0, stream, x, w4, bias, residual, out, pre, M, N, K, am_x, am_w, am_out)
        if (np == 3) { if (narrow) VIT_LAUNCH_X6S(1, 3); else VIT_LAUNCH_X6S(2, 3); }
        else if (np == 2) { if (narrow) VIT_LAUNCH_X6S(1, 2); else VIT_LAUNCH_X6S(2, 2); }
        else { if (narrow) VIT_LAUNCH_X6S(1, 6); else VIT_LAUNCH_X6S(2, 6); }
#undef VIT_LAUNCH_X6S
    } else {
#define VIT_LAUNCH_X6(ACT, TN)                                                                                                                          \
    do {                                                                                                                                                \
        if (np == 3) hipLaunchKernelGGL((x6::k_linear_x6<ACT, TN, false, 3>), dim3(tiles), dim3(256), 0, stream, x, w4, bias, residual, out, pre, M, N, K, am_x, am_w, am_out); \
        else if (np == 2) hipLaunchKernelGGL((x6::k_linear_x6<ACT, TN, false, 2>), dim3(tiles), dim3(256), 0, stream, x, w4, bias, residual, out, pre, M, N, K, am_x, am_w, am_out); \
        else hipLaunchKernelGGL((x6::k_linear_x6<ACT, TN, false, 6>), dim3(tiles), dim3(256), 0, stream, x, w4, bias, residual, out, pre, M, N, K, am_x, am_w, am_out);      \
    } while (0)
        if (act == 1) { if (narrow) VIT_LAUNCH_X6(1, 1); else VIT_LAUNCH_X6(1, 2); }
        else if (act == 2) { if (narrow) VIT_LAUNCH_X6(2, 1); else VIT_LAUNCH_X6(2, 2); }
        else { if (narrow) VIT_LAUNCH_X6(0, 1); else VIT_LAUNCH_X6(0, 2); }
#undef VIT_LAUNCH_X6
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    if (am_out && S > 1) return amax(out, (int64_t)M * N, am_out, stream);      // split-K partial sums: the |max| of the result needs its own pass
    return VIT_OK;
}

// Contraction splits of the Linear weight gradient from a measured cost model (tools/probes/wgrad_lab.py, VIT_WGRAD_S sweeps at M = 5 140,
// profiles/r03_wgrad_splits.md).  k_wgrad_x6 keeps up to three workgroups resident per CU; a workgroup advances one 16-row slab every
// t[c] us with c of them on its CU, and the CU's throughput saturates at two (t_sat = t[2] / 2); every workgroup also pays ~8 slabs' worth
// of pipeline fill + atomic epilogue, every split one more pass of atomics over the output:
//     cost(S) = max(chain x t[c], workgroups per CU x chain x t_sat) + S us,    chain = slabs per split + 8
// r02's rule ("fill whole rounds of 512 slots") put the 192- / 108- / 144-tile outputs (qkv, decoder fc) at S = 8 / 14 / 7 where S = 4 / 7 / 5 is 15 - 25 % faster.
static int wgrad_splits(int tiles, int nslab, int products)
{
    static const float t3[4] = {0.f, 0.66f, 1.07f, 1.53f}, t6[4] = {0.f, 0.95f, 1.62f, 2.48f};
    const float *t = products == 3 ? t3 : t6;
    const float t_sat = 0.5f * t[2];
    int best = 1;
    float best_cost = 1e30f;
    for (int S = 1; S <= 64; ++S) {
        if (S > 1 && nslab / S < 8) break;
        const float per_cu = (float)tiles * (float)S / 256.f, chain = (float)((nslab + S - 1) / S) + 8.f;
        const int c = per_cu <= 1.f ? 1 : (per_cu <= 2.f ? 2 : 3);
        const float cost = fmaxf(chain * t[c], per_cu * chain * t_sat) + (float)S;
        if (cost < best_cost) { best_cost = cost; best = S; }
    }
    return best;
}

int linear_x6_wgrad(const float *dy, const float *x, float *dw, float *dbias, int M, int N, int K, int accumulate,
                    hipStream_t stream)
{
    if (!dy || !x || !dw || M <= 0 || N <= 0 || K <= 0) return VIT_EINVAL;
    const int tiles = ((N + x6::BM - 1) / x6::BM) * ((K + 127) / 128);
    const int nslab = (M + x6::BK - 1) / x6::BK;
    // split M so that tiles x S fills the 512 resident workgroups (256 CUs x 2) ONCE: 1.1 rounds cost as much as 2
    const int np = x6_products();
    const uint32_t *am_dy, *am_x;
    take_amax(am_dy, am_x);
    if (np == 2 && (!am_dy || !am_x)) return VIT_EINVAL;
    int S = wgrad_splits(tiles, nslab, np == 6 ? 6 : 3);
    { static const char *force = getenv("VIT_WGRAD_S"); if (force) S = atoi(force); }   // (tools/probes/wgrad_lab.py sweeps)
    // the kernel addresses a split's rows with 32-bit byte offsets from the split's first row (+ 4 slabs of prefetch)
    while (((int64_t)(nslab / S) + 6) * x6::BK * (int64_t)(N > K ? N : K) * 4 >= 0x7fffffffLL && S < 65535) ++S;
    if (((int64_t)(nslab / S) + 6) * x6::BK * (int64_t)(N > K ? N : K) * 4 >= 0x7fffffffLL) return VIT_EINVAL;
    (void)hipGetLastError();
    if (!accumulate) {   // accumulate: dw / dbias already hold the running gradient (e.g. a zeroed all-reduce bucket slice)
        // one memset when the caller laid db out right behind dw (vit_ops.py does)
        const bool joined = dbias == dw + (size_t)N * K;
        if (S > 1 && hipMemsetAsync(dw, 0, ((size_t)N * K + (joined ? N : 0)) * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
        if (dbias && !(S > 1 && joined) && hipMemsetAsync(dbias, 0, (size_t)N * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
    }
    if (np == 3) hipLaunchKernelGGL(x6::k_wgrad_x6<3>, dim3(tiles, S), dim3(256), 0, stream, dy, x, dw, dbias, M, N, K, accumulate, am_dy, am_x);
    else if (np == 2) hipLaunchKernelGGL(x6::k_wgrad_x6<2>, dim3(tiles, S), dim3(256), 0, stream, dy, x, dw, dbias, M, N, K, accumulate, am_dy, am_x);
    else hipLaunchKernelGGL(x6::k_wgrad_x6<6>, dim3(tiles, S), dim3(256), 0, stream, dy, x, dw, dbias, M, N, K, accumulate, am_dy, am_x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
int conv_x6_fwd(const float *in, const void *wp, const float *bias, const float *residual, float *out, int B, int Ci, int Co,
                int H, int W, int ksize, int flags, hipStream_t stream)
{
    const int relu_in = flags & 1, gate = (flags >> 1) & 1;
    if (gate && (!residual || bias)) return VIT_EINVAL;
    if (!in || !wp || !out || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return VIT_EINVAL;
    if ((ksize != 1 && ksize != 3) || (Ci % x6::BK) != 0) return VIT_EINVAL;
    const int64_t NP = (int64_t)B * H * W;
    const int64_t tiles = (int64_t)((Co + x6::BM - 1) / x6::BM) * ((NP + 127) / 128);
    if (tiles > 0x7fffffff) return VIT_EINVAL;
    const uint4 *w4 = static_cast<const uint4 *>(wp);
    const int np = x6_products();
    const uint32_t *am_in, *am_unused, *am_w = weight_amax(wp, Co, ksize * ksize * Ci);
    take_amax(am_in, am_unused);
    uint32_t *am_out = x6_take_output_amax();
    if (np == 2 && !am_in) return VIT_EINVAL;
    (void)hipGetLastError();
    // 3x3 layers at least one 32-pixel patch row wide: the halo kernel (one fetch / split per input element instead of nine)
    static const bool taps_only = [] { const char *e = getenv("VIT_CONV3"); return e && e[0] == 't'; }();      // VIT_CONV3=taps: A/B switch
    if (ksize == 3 && W >= 32 && !taps_only) {
        const int64_t ht = (int64_t)((Co + x6::BM - 1) / x6::BM) * B * ((W + 31) / 32) * ((H + 3) / 4);
        if (ht > 0x7fffffff) return VIT_EINVAL;
        int S = 1;
        const int nslab = Ci / x6::BK;
        if (ht < 256) { S = (int)(512 / ht); while (S > 1 && nslab / S < 2) --S; if (S < 1) S = 1; if (S > 8) S = 8; }
        if (S > 1 && !zero_fill(out, (size_t)NP * Co * sizeof(float), stream)) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
#define VIT_LAUNCH_H6(RL)                                                                                                                                      \
    do {                                                                                                                                                        \
        if (np == 3) hipLaunchKernelGGL((x6::k_conv3h_x6<RL, 3>), dim3((unsigned)ht, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in, am_out); \
        else if (np == 2) hipLaunchKernelGGL((x6::k_conv3h_x6<RL, 2>), dim3((unsigned)ht, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in, am_out); \
        else hipLaunchKernelGGL((x6::k_conv3h_x6<RL, 6>), dim3((unsigned)ht, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in, am_out);      \
    } while (0)
        if (relu_in) VIT_LAUNCH_H6(true); else VIT_LAUNCH_H6(false);
#undef VIT_LAUNCH_H6
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
        if (am_out && S > 1) return amax(out, NP * Co, am_out, stream);      // partial sums: the |max| of the result needs its own pass
        return VIT_OK;
    }
    // few output tiles: split the K slabs so that tiles x S fills the 512 resident workgroups, >= 8 slabs per split
    int S = 1;
    const int nslab = ksize * ksize * Ci / x6::BK;
    if (tiles < 256) { S = (int)(512 / tiles); while (S > 1 && nslab / S < 8) --S; if (S < 1) S = 1; if (S > 16) S = 16; }
    if (S > 1 && !zero_fill(out, (size_t)NP * Co * sizeof(float), stream)) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
#define VIT_LAUNCH_C6(KS, RL)                                                                                                                                   \
    do {                                                                                                                                                        \
        if (np == 3) hipLaunchKernelGGL((x6::k_conv_x6<KS, RL, 3>), dim3((unsigned)tiles, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in); \
        else if (np == 2) hipLaunchKernelGGL((x6::k_conv_x6<KS, RL, 2>), dim3((unsigned)tiles, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in); \
        else hipLaunchKernelGGL((x6::k_conv_x6<KS, RL, 6>), dim3((unsigned)tiles, S), dim3(256), 0, stream, in, w4, bias, residual, out, B, Ci, Co, H, W, gate, am_w, am_in);      \
    } while (0)
    if (ksize == 3) { if (relu_in) VIT_LAUNCH_C6(3, true); else VIT_LAUNCH_C6(3, false); }
    else { if (relu_in) VIT_LAUNCH_C6(1, true); else VIT_LAUNCH_C6(1, false); }
#undef VIT_LAUNCH_C6
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    if (am_out) return amax(out, NP * Co, am_out, stream);      // (k_conv_x6 does not publish its output's |max| itself: one pass here)
    return VIT_OK;
}
int conv_x6_wgrad(const float *dy, const float *in, float *dw, float *dbias, int B, int Ci, int Co, int H, int W, int ksize,
                  int relu_in, hipStream_t stream)
{
    if (!dy || !in || !dw || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return VIT_EINVAL;
    if ((ksize != 1 && ksize != 3) || (W % 8) != 0 || ((H * W) % x6::BK) != 0 || (int64_t)B * Ci * H * W < 8) return VIT_EINVAL;
    const int R = ksize * ksize * Ci;
    const int tiles = ((Co + x6::BM - 1) / x6::BM) * ((R + 127) / 128);
    const int nslab = B * (H * W / x6::BK);
    int S = split_count(tiles, nslab, 32);   // (r03: 8 / 4 slabs per split measured slower at every DPT shape -- more atomic passes over the tile)
    const int np = x6_products();
    const uint32_t *am_dy, *am_in;
    take_amax(am_dy, am_in);
    if (np == 2 && (!am_dy || !am_in)) return VIT_EINVAL;
    (void)hipGetLastError();
    if (hipMemsetAsync(dw, 0, (size_t)Co * R * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
    if (dbias && hipMemsetAsync(dbias, 0, (size_t)Co * sizeof(float), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
#define VIT_LAUNCH_G6(KS, RL)                                                                                                                          \
    do {                                                                                                                                               \
        if (np == 3) hipLaunchKernelGGL((x6::k_conv_wgrad_x6<KS, RL, 3>), dim3(tiles, S), dim3(256), 0, stream, dy, in, dw, dbias, B, Ci, Co, H, W, am_dy, am_in);      \
        else if (np == 2) hipLaunchKernelGGL((x6::k_conv_wgrad_x6<KS, RL, 2>), dim3(tiles, S), dim3(256), 0, stream, dy, in, dw, dbias, B, Ci, Co, H, W, am_dy, am_in);   \
        else hipLaunchKernelGGL((x6::k_conv_wgrad_x6<KS, RL, 6>), dim3(tiles, S), dim3(256), 0, stream, dy, in, dw, dbias, B, Ci, Co, H, W, am_dy, am_in);            \
    } while (0)
    if (ksize == 3) { if (relu_in) VIT_LAUNCH_G6(3, true); else VIT_LAUNCH_G6(3, false); }
    else { if (relu_in) VIT_LAUNCH_G6(1, true); else VIT_LAUNCH_G6(1, false); }
#undef VIT_LAUNCH_G6
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
