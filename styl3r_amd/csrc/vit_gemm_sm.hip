// vit_gemm_sm.hip -- the split-arithmetic Linear for SMALL row counts (M <= ~1 000: batch-1 serving, C2: 257 / 514 token rows) and, at any row count,
// for NARROW outputs (N <= 768: the decoders' proj / projq / projk / projv / fc2 and the input-gradient GEMMs of their qkv / fc1 in the train step).
// Reached through vit_linear_x6r_fwd with cfg = 5 (one problem) and vit_linear_sm_grouped (two problems of one shape: the dual decoders).
//
//     out (M,N) = [residual +] act( x (M,K) . w^T (N,K) + bias )            same operands, split functions and epilogue as vit_gemm_x6.hip; BLOCK weight image
//
// What is different from k_linear_x6 (128-row tiles, LDS-staged operands, a barrier pair per 16-deep slab, split-K through fp32 atomics into a
// zero-filled output when the tiles cannot fill the chip -- at M = 257 / 514 that is a zero-fill launch, 640 workgroups of 8 slabs each, 128-row
// tiles whose last one holds 1 or 2 rows, no fused activation, no |max| word):
//   * NO barrier in the main loop: the contraction is split over the NW waves of a workgroup (wave w takes the w-th range of 32-deep stages
//     of the SAME 32 TM x 64 output tile) and every wave runs its own pipeline on its own operands:
//       - the WEIGHT pieces come straight from global memory into MFMA registers, from the BLOCK image (vit_split_weight_block:
//         packed[n / 64][k / 8][piece][n % 64][8]): the 32 lanes of a half wave read 512 consecutive bytes -- the B operand of
//         v_mfma_f32_32x32x16 as it lies in memory.  (A first version read the row image, lane = weight row: 64 cache lines per wave load,
//         2 960 cycles per 16-deep slab -- bound by the address rate of the texture path, 8 x slower than its MFMAs;
//         profiles/r06_small_linear_lab_v1.jsonl.)
//       - the ACTIVATIONS are loaded row-contiguously (8 lanes x 16 B = one 128-byte line of one row), split in registers and transposed
//         through a wave-private double buffer in LDS (no barrier: the LDS queue of a wave is in order), two stages ahead.
//   * the partial tiles meet in LDS once, are added in wave order (deterministic) and the full epilogue -- bias, GELU / GELU', residual,
//     `pre`, the |max| word of the stored values -- runs on row-major, coalesced stores.  No zero fill, no atomics, no second pass.
//   * 64- or 32-row tiles: M = 514 is 9 x 64 rows (576) instead of 5 x 128 (640), M = 257 is 9 x 32 (288) instead of 3 x 128 (384).
// Reference shapes: croco/blocks.py:76-82 (Mlp), :97-134 (Attention qkv / proj), :171-200 (CrossAttention projq / projk / projv) at the row counts of
// infer_model_re10k.py:262-560 (two context views + one style image at batch 1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"
#include "vit_amax.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;
int x6_products();
void x6_take_amax(const uint32_t *&a, const uint32_t *&b);
uint32_t *x6_take_output_amax();

namespace sm {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ inline float gelu_grad_exact(float x)
{
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// two fp32 values -> their three bf16 pieces (the functions of vit_gemm_x6.hip: a tensor split here and there gets the same pieces)
__device__ inline void split2_bf16(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    f32x2 f = {a, b};
    const bf16x2 h0 = __builtin_convertvector(f, bf16x2);
    const f32x2 r1 = f - __builtin_convertvector(h0, f32x2);
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    p0 = __builtin_bit_cast(uint32_t, h0); p1 = __builtin_bit_cast(uint32_t, h1); p2 = __builtin_bit_cast(uint32_t, h2);
}
template <int NPROD> __device__ inline f32x16 mma(const uint4 &a, const uint4 &b, const f32x16 &c)
{
    if (NPROD == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int LPAD = 4;          // floats of padding behind a column of the partial tile in LDS (column stride 32 TM + 4: b128 accesses of 8 lanes hit 32 banks)

// compiler fence: memory accesses (global loads, LDS traffic) stay on their side.  The MFMAs of a stage read their A pieces from LDS, so they are
// pinned with them; without the fences the optimiser sinks every global load to its first use -- the opposite of a prefetch.
__device__ inline void fence() { asm volatile("" ::: "memory"); }

// LDS image of one activation stage (32 k) of one wave: [piece][row][4 k groups x 16 B]; k group g of row r lives in slot g ^ ((r >> 2) & 3): the 16
// lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}: MI355X_MICROARCH LDS table) then hit 16 distinct 4-bank slots
__device__ inline int a_slot(int row, int g) { return row * 4 + (g ^ ((row >> 2) & 3)); }

// TM: 32-row MFMA blocks per tile (tile = 32 TM rows x 64 columns = one block row of the weight image); NW: waves per workgroup = contraction
// ranges; NPROD: 6 / 3 (bf16 pieces) or 2 (fp16 pieces).  act: 0 none, 1 exact GELU, 2 "times GELU'(residual)" (the input-gradient GEMM behind
// a GELU: `residual` carries the pre-activation).  Requires N % 64 == 0 and K % (64 NW) == 0 (an even number of 32-deep stages per wave).
// Up to two problems of one shape per launch (blockIdx.y): the dual decoders of the serving path run the same layer with two weight sets
// (vit_linear_sm_grouped; backbone_croco_multiview.py:147-188 at two context views).  The operands of a group are its own; the |max| words of
// the activations are shared (the stacked input / output tensor is one tensor with one scale).
struct SmGroup { const float *x; const uint4 *wpb; const float *bias; const float *residual; float *out; float *pre; const uint32_t *amax_w; };
struct SmArgs { SmGroup g[2]; int M, N, K, act; const uint32_t *amax_x; uint32_t *amax_out; };

template <int TM, int NW, int NPROD>
__global__ void __launch_bounds__(NW * 64) k_linear_sm(const SmArgs args)
{
    const SmGroup &grp = args.g[blockIdx.y];
    const float *x = grp.x;
    const uint4 *wpb = grp.wpb;
    const float *__restrict__ bias = grp.bias, *__restrict__ residual = grp.residual;
    float *__restrict__ out = grp.out, *__restrict__ pre = grp.pre;
    const int M = args.M, N = args.N, K = args.K, act = args.act;
    const uint32_t *__restrict__ amax_x = args.amax_x, *__restrict__ amax_w = grp.amax_w;
    uint32_t *__restrict__ amax_out = args.amax_out;
    constexpr int BMT = 32 * TM, BNT = 64, NPC = NPROD == 6 ? 3 : 2;
    constexpr int NLA = BMT / 8;                       // activation loads per stage and lane (8 rows x 128 B per wave instruction)
    constexpr int ABUF = NPC * BMT * 4;                // uint4 slots of one LDS stage image
    constexpr int CS = BMT + LPAD;                     // column stride of a partial tile in LDS (floats)
    extern __shared__ uint4 lds[];                     // main loop: [wave][2][ABUF]; afterwards the partial tiles [wave][64][CS] floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    // workgroup id -> tile: consecutive ids are dealt round-robin to the 8 XCDs; XCD x takes one contiguous range of the tile sequence, which
    // walks the row tiles of one weight block row first: the (few) row tiles that share it run side by side on one L2
    const int tiles_m = (M + BMT - 1) / BMT, tiles_n = N / BNT, ntiles = tiles_m * tiles_n;
    const int q = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7;
    const int pid = xcd * q + min(xcd, r8) + (blockIdx.x >> 3);
    const int tm = pid % tiles_m, rb = pid / tiles_m;
    const int m0 = tm * BMT, n0 = rb * BNT;
    const int ns = (K >> 5) / NW;                      // 32-deep stages of this wave (even, >= 2: the host checks)
    const int k_lo = wave * ns * 32;
    // activation loader: load t of a stage = rows 8 t + (lane >> 3), 16-byte segment lane & 7 of the row's 128 bytes; rows past M are CLAMPED
    // (their products land in accumulator rows the epilogue never stores)
    const int lrow = lane >> 3, seg = lane & 7;
    int aoff[NLA];
#pragma unroll
    for (int t = 0; t < NLA; ++t) aoff[t] = min(m0 + 8 * t + lrow, M - 1) * K + k_lo + seg * 4;
    // weight loader: (slab s, column half j, piece c) of a stage = uint4 ((rb KG + kg) 3 + c) 64 + 32 j + col, kg = 4 stage + 2 s + half
    const int KG = K >> 3;
    const int boff = ((rb * KG + (k_lo >> 3) + half) * 3) * 64 + col;
    struct RA { float4 v[NLA]; };
    struct RB { uint4 v[2][2][NPC]; };
    auto load_a = [&](RA &ra, int st) {
        const int st_ = min(st, ns - 1);               // past the range: the last stage again (never consumed)
#pragma unroll
        for (int t = 0; t < NLA; ++t) ra.v[t] = *reinterpret_cast<const float4 *>(x + aoff[t] + st_ * 32);
    };
    auto load_b = [&](RB &rb_, int st) {
        const int st_ = min(st, ns - 1);
        const uint4 *p = wpb + boff + st_ * (4 * 3 * 64);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < NPC; ++c) rb_.v[s][j][c] = p[(2 * s * 3 + c) * 64 + 32 * j];
    };
    float sx = 1.f, ix = 1.f, iw = 1.f;
    uint4 *ring = lds + (size_t)wave * 2 * ABUF;
    // split one stage of raw activations and store its pieces: lane (row, seg) holds k = 4 seg .. 4 seg + 3 -> half a k group of every piece
    auto write_a = [&](int buf, const RA &ra) {
        uint2 *base = reinterpret_cast<uint2 *>(ring + buf * ABUF);
#pragma unroll
        for (int t = 0; t < NLA; ++t) {
            const int row = 8 * t + lrow;
            const float4 v = ra.v[t];
            uint32_t p0[2], p1[2], p2[2];
            if (NPROD == 2) {
                f16_split2(v.x * sx, v.y * sx, p0[0], p1[0]);
                f16_split2(v.z * sx, v.w * sx, p0[1], p1[1]);
            } else {
                split2_bf16(v.x, v.y, p0[0], p1[0], p2[0]);
                split2_bf16(v.z, v.w, p0[1], p1[1], p2[1]);
            }
            const int slot2 = a_slot(row, seg >> 1) * 2 + (seg & 1);          // in uint2 units
            base[slot2] = make_uint2(p0[0], p0[1]);
            base[BMT * 8 + slot2] = make_uint2(p1[0], p1[1]);
            if (NPC == 3) base[2 * BMT * 8 + slot2] = make_uint2(p2[0], p2[1]);
        }
    };
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
    auto compute = [&](int buf, const RB &rb_) {
        const uint4 *a = ring + buf * ABUF;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 pa[TM][NPC];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int c = 0; c < NPC; ++c) pa[i][c] = a[c * BMT * 4 + a_slot(32 * i + col, 2 * s + half)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];           // smallest partial products first (the order of k_linear_x6)
                    if (NPROD == 6) {
                        c = mma<NPROD>(pa[i][NPC - 1], rb_.v[s][j][0], c);
                        c = mma<NPROD>(pa[i][1], rb_.v[s][j][1], c);
                        c = mma<NPROD>(pa[i][0], rb_.v[s][j][NPC - 1], c);
                    }
                    c = mma<NPROD>(pa[i][1], rb_.v[s][j][0], c);
                    c = mma<NPROD>(pa[i][0], rb_.v[s][j][1], c);
                    c = mma<NPROD>(pa[i][0], rb_.v[s][j][0], c);
                    acc[i][j] = c;
                }
        }
    };
    // Pipeline (two register sets P / Q, two LDS images 0 / 1): stage st computes from image st & 1 and weight set st & 1 while the weight loads
    // of stage st + 2, the raw activations of stages st + 2 / st + 3 and the image of stage st + 1 are on their way.
    RA ra_p, ra_q;
    RB rb_p, rb_q;
    load_a(ra_p, 0); load_b(rb_p, 0); load_a(ra_q, 1); load_b(rb_q, 1);
    fence();
    // f16x3: the operand scales, read BEHIND the first operand loads (two more round trips to memory that would otherwise sit in front of them)
    if (NPROD == 2) { sx = f16_scale_of(amax_word_read(amax_x)); ix = 1.f / sx; iw = 1.f / f16_scale_of(amax_word_read(amax_w)); }
    fence();
    write_a(0, ra_p);
    fence();
    load_a(ra_p, 2);
    fence();
    for (int st = 0; st < ns; st += 2) {
        compute(0, rb_p);
        fence();
        load_b(rb_p, st + 2);
        fence();
        write_a(1, ra_q);
        fence();
        load_a(ra_q, st + 3);
        fence();
        compute(1, rb_q);
        fence();
        load_b(rb_q, st + 3);
        fence();
        write_a(0, ra_p);
        fence();
        load_a(ra_p, st + 4);
        fence();
    }
    // the epilogue's global operands, requested before the two barriers of the reduction: thread = (column c = lane, row group = wave)
    constexpr int RPT = BMT / NW;                        // rows per thread (a multiple of 4 for every instantiated shape)
    static_assert(RPT % 4 == 0 && RPT * NW == BMT, "tile / workgroup shape");
    const int c = lane, rg = wave, n = n0 + c;
    const float bv = bias ? bias[n] : 0.f;
    float resv[RPT];
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
        const int m = m0 + rg * RPT + e;
        resv[e] = (residual && m < M) ? residual[(int64_t)m * N + n] : 0.f;
    }
    fence();
    mfma_result_fence();
    __syncthreads();                                   // every wave is done with its ring: the partial tiles take the memory
    float *part = reinterpret_cast<float *>(lds);
    // partial tiles -> LDS, column-major: register r of block (i, j) is row 32 i + (r & 3) + 8 (r >> 2) + 4 half of column 32 j + col:
    // four consecutive registers are four consecutive rows -> one ds_write_b128
    float *mine = part + (size_t)wave * BNT * CS;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                *reinterpret_cast<float4 *>(mine + (32 * j + col) * CS + 32 * i + 8 * g + 4 * half) = v;
            }
    __syncthreads();
    // epilogue: rows rg * RPT .. + RPT of column c per thread, four rows per ds_read_b128; the 64 lanes of a wave store 64 consecutive floats of
    // one output row
    uint32_t omax = 0;
#pragma unroll
    for (int t = 0; t < RPT / 4; ++t) {
        const int row = rg * RPT + 4 * t;
        float4 sum = *reinterpret_cast<const float4 *>(part + c * CS + row);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(part + ((size_t)w * BNT + c) * CS + row);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const float vals[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + row + e;
            if (m >= M) continue;
            const int64_t o = (int64_t)m * N + n;
            float v = vals[e];
            if (NPROD == 2) v = v * ix * iw;
            v += bv;
            if (act == 2) {
                v *= gelu_grad_exact(resv[4 * t + e]);
            } else {
                if (pre) pre[o] = v;
                if (act == 1) v = gelu_exact(v);
                if (residual) v += resv[4 * t + e];
            }
            out[o] = v;
            omax = max(omax, abs_bits(v));
        }
    }
    if (amax_out) amax_word_fold(amax_out, omax);
}

struct Cfg { int tm, nw; };
// Tile rows and waves for a shape (0: no configuration fits -- the caller keeps the 128-row kernel).  Measured on the C2 shapes with
// tools/probes/small_linear_lab.py (profiles/r06_small_linear_lab.md).
static bool fits(int K, int nw) { return (K % (64 * nw)) == 0; }
// (six products, 64-row tiles, 8 waves: the rings of three-piece images would need 196 KB of LDS -- not instantiated)
static bool built(int tm, int nw, int np) { return !(np == 6 && tm == 2 && nw == 8); }
static Cfg choose(int M, int N, int K)
{
    // profiles/r06_small_linear_lab.jsonl (f16x3 / bf16x6, C2 shapes, weights from HBM): 64-row tiles pay when they alone fill the chip
    // (>= 400 tiles) or when the contraction is long (K >= 3072: half the weight re-reads); 32-row tiles otherwise (M = 257 is 9 x 32 = 288 rows,
    // not 5 x 64 = 320) and always in six-product mode (the 64-row kernel holds one wave per SIMD there); eight waves when a long contraction
    // meets a tile count that leaves CUs empty.
    Cfg c;
    const int np = x6_products();
    const int waste64 = ((M + 63) / 64) * 64 - M, waste32 = ((M + 31) / 32) * 32 - M;
    const int tiles64 = ((M + 63) / 64) * (N / 64);
    if (M <= 32 || np == 6 || (waste64 - waste32) * 10 > M) c.tm = 1;
    else c.tm = (tiles64 >= 400 || K >= 3072) ? 2 : 1;
    const int tiles = ((M + 32 * c.tm - 1) / (32 * c.tm)) * (N / 64);
    c.nw = (K >= 3072 && tiles < 300 && fits(K, 8) && built(c.tm, 8, np)) ? 8 : 4;
    if (!fits(K, c.nw)) c.nw = 0;
    return c;
}
static thread_local int g_force_tm = 0, g_force_nw = 0, g_max_rows = 1024;
template <int TM, int NW, int NPROD> constexpr size_t lds_bytes()
{
    constexpr size_t ring = (size_t)NW * 2 * (NPROD == 6 ? 3 : 2) * 32 * TM * 64, part = (size_t)NW * 64 * (32 * TM + LPAD) * sizeof(float);
    return ring > part ? ring : part;
}
}  // namespace sm

// vit_linear_sm_set: launches of vit_linear_x6r_fwd with cfg 5 and M <= max_rows take the small-M kernel (0: never -- the A/B switch of tools/
// and tests); tm / nw force the tile rows (1 / 2 blocks of 32) and the waves per workgroup (4 / 8) of the next launches, 0 = the rule of
// sm::choose.  Per host thread, like vit_x6_set_products.
int linear_sm_set(int max_rows, int tm, int nw)
{
    if (max_rows < 0 || (tm != 0 && tm != 1 && tm != 2) || (nw != 0 && nw != 4 && nw != 8)) return VIT_EINVAL;
    sm::g_max_rows = max_rows; sm::g_force_tm = tm; sm::g_force_nw = nw;
    return VIT_OK;
}

// 1 when (M, N, K) runs on the small-M kernel under the current vit_linear_sm_set state
int linear_sm_ok(int M, int N, int K)
{
    if (M < 1 || M > sm::g_max_rows || (N % 64) != 0 || K < 128 || (int64_t)M * K >= (1ll << 31) || (int64_t)N * K >= (1ll << 31)) return 0;   // (32-bit element offsets)
    const sm::Cfg c = sm::choose(M, N, K);
    const int nw = sm::g_force_nw ? sm::g_force_nw : c.nw, tm = sm::g_force_tm ? sm::g_force_tm : c.tm;
    return nw != 0 && sm::fits(K, nw) && sm::built(tm, nw, x6_products());
}

static int launch_sm(const sm::SmArgs &a, int groups, hipStream_t stream)
{
    const int np = x6_products(), M = a.M, N = a.N, K = a.K;
    if (np == 2 && !a.amax_x) return VIT_EINVAL;
    if (!linear_sm_ok(M, N, K)) return VIT_EINVAL;
    sm::Cfg c = sm::choose(M, N, K);
    if (sm::g_force_tm) c.tm = sm::g_force_tm;
    if (sm::g_force_nw) c.nw = sm::g_force_nw;
    const int tiles = ((M + 32 * c.tm - 1) / (32 * c.tm)) * (N / 64);
    (void)hipGetLastError();
#define VIT_SM_LAUNCH(TM, NW, NP)                                                                                                                 \
    do {                                                                                                                                         \
        auto kern = sm::k_linear_sm<TM, NW, NP>;                                                                                                 \
        constexpr size_t lds = sm::lds_bytes<TM, NW, NP>();                                                                                      \
        static bool attr_set = false;    /* (idempotent: a race sets it twice) */                                                               \
        if (!attr_set && lds > 64 * 1024) {                                                                                                      \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
                g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH;                                                                        \
            }                                                                                                                                    \
            attr_set = true;                                                                                                                     \
        }                                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(tiles, groups), dim3(NW * 64), lds, stream, a);                                                           \
    } while (0)
#define VIT_SM_NP(TM, NW)                                                                                                                         \
    do {                                                                                                                                         \
        if (np == 2) VIT_SM_LAUNCH(TM, NW, 2); else if (np == 3) VIT_SM_LAUNCH(TM, NW, 3); else VIT_SM_LAUNCH(TM, NW, 6);                         \
    } while (0)
    if (c.tm == 1) { if (c.nw == 8) VIT_SM_NP(1, 8); else VIT_SM_NP(1, 4); }
    else if (c.nw == 8) { if (np == 2) VIT_SM_LAUNCH(2, 8, 2); else VIT_SM_LAUNCH(2, 8, 3); }
    else VIT_SM_NP(2, 4);
#undef VIT_SM_NP
#undef VIT_SM_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

static const uint32_t *block_image_amax(const void *wpb, int N, int K)
{
    return reinterpret_cast<const uint32_t *>(static_cast<const char *>(wpb) + (size_t)((N + 63) / 64) * 64 * (size_t)K * 6);
}

// wpb: the BLOCK image of the weight (vit_split_weight_block)
int linear_sm_fwd(const float *x, const void *wpb, const float *bias, const float *residual, float *out, float *pre, int M, int N, int K, int act,
                  const uint32_t *am_x, const uint32_t *am_w, uint32_t *am_out, hipStream_t stream)
{
    sm::SmArgs a{};
    a.g[0] = sm::SmGroup{x, static_cast<const uint4 *>(wpb), bias, residual, out, pre, am_w};
    a.g[1] = a.g[0];
    a.M = M; a.N = N; a.K = K; a.act = act; a.amax_x = am_x; a.amax_out = am_out;
    return launch_sm(a, 1, stream);
}

// Two Linear layers of one shape in one launch (see sm::SmGroup): out_g = [residual_g +] act(x_g . w_g^T + bias_g), g = 0, 1.  The announced
// activation |max| word (vit_x6_set_operand_amax) covers BOTH inputs, the output word (vit_x6_set_output_amax) both outputs.
int linear_sm_grouped(const float *const *x, const void *const *wpb, const float *const *bias, const float *const *residual, float *const *out,
                      int groups, int M, int N, int K, int act, hipStream_t stream)
{
    if (!x || !wpb || !out || groups < 1 || groups > 2 || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return VIT_EINVAL;
    const uint32_t *ax, *unused;
    x6_take_amax(ax, unused);
    sm::SmArgs a{};
    for (int g = 0; g < groups; ++g) {
        if (!x[g] || !wpb[g] || !out[g] || (act == 2 && !(residual && residual[g]))) return VIT_EINVAL;
        a.g[g] = sm::SmGroup{x[g], static_cast<const uint4 *>(wpb[g]), bias ? bias[g] : nullptr, residual ? residual[g] : nullptr, out[g], nullptr,
                             block_image_amax(wpb[g], N, K)};
    }
    if (groups == 1) a.g[1] = a.g[0];
    a.M = M; a.N = N; a.K = K; a.act = act; a.amax_x = ax; a.amax_out = x6_take_output_amax();
    return launch_sm(a, groups, stream);
}
}  // namespace vit
