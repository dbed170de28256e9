// vit_gemm_x6r.hip -- the bf16x6 Linear (see vit_gemm_x6.hip for the arithmetic) with an LDS-DMA operand ring.
//
//     out (M,N) = [residual +] act( x (M,K) . w^T (N,K) + bias )
//
// EXPERIMENTAL: exported (vit_linear_x6r_fwd) and parity-tested, not on the default path of styl3r_amd/vit_ops.py.
// Round 2's answer to "is vit_gemm_x6.hip's register-staged data path what holds it at ~40 % of the matrix peak?"
// Measured on the encoder's qkv Linear (5140 x 3072 x 1024, random operands, warm clocks, one run; profiles/r02s_gemm_lab.jsonl,
// DESIGN 9.2 has every shape):
//     vit_gemm_x6.hip, 128 x 128 tiles, registers -> split -> ds_write, 3 workgroups / CU          170 TF
//     k_linear_x6r, 128 x 128 ring, split at fragment read, 2 workgroups / CU            (cfg 1)   177 TF
//     k_linear_x6r, 256 x 256 ring, 8 waves in lockstep                                  (cfg 2)   185 TF
//     k_linear_x6c, 256 x 256, split once per workgroup, wave pairs in ping-pong          (cfg 3)   206 TF   (280 TF on zero-filled operands)
// The data path alone is worth +4 %; what moves the number is the ping-pong schedule: 3 460 cycles per 16-wide slab against
// the 3 072 its 2 x 48 MFMAs per SIMD need (lockstep: 5 000).  At that point the kernel is limited by the power budget, not by
// cycles: the same binary on zero-filled operands runs the same cycle counts at 2.3 GHz instead of 1.65 GHz.  256 x 256 tiles
// quantise badly on everything but the qkv / fc1 shapes (84 tiles for N = 1024) and a K split (vit_linear_x6c_fwd: partial tiles
// through a workspace, last arriver reduces) only pays for K >= 3072, so the default path keeps the 128-wide kernel.
//
// What this file does differently:
//   * both operands reach LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no LDS store instructions;
//     every DMA instruction moves 1 KiB, each 4-lane group reading 64 contiguous bytes;
//   * the weight is pre-split in a BLOCK layout, packed[n / 64][k / 8][piece][n % 64][8] bf16 (vit_split_weight_block):
//     a DMA chunk is 1 KiB of contiguous global memory and lands as one plane of 64 consecutive 16-byte LDS slots, which
//     is also the conflict-free order for the B fragment reads;
//   * k_linear_x6r: the activations stay fp32 in LDS and are split into their three bf16 pieces when a wave reads its
//     fragment (redundantly in the WN waves that share the rows); NST stages form a ring, the DMA for slab s + NST - 1 is
//     issued when slab s starts and waited for with a COUNTED `s_waitcnt vmcnt(n)`, one raw `s_barrier` per slab;
//   * k_linear_x6c: every wave converts a 32-row share of the NEXT slab into bf16 planes (each activation is split once
//     per workgroup), all fragment reads are plane reads, and the two waves that share a SIMD run their LDS phase and
//     their MFMA phase in opposite order, so one wave's 48 MFMAs cover the other's reads and conversion.
//
// LDS images: raw A = BM rows x 64 B, 16-byte slot q of row r at slot r*4 + (q ^ ((r >> 2) & 3)) (the 16 lanes of every
// ds_read_b128 lane group hit 16 distinct 4-bank slots; the DMA applies the XOR on its SOURCE address because its
// destination is lane-linear); planes = [k group 2][piece 3][rows][16 B].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_ops.h"

namespace vit {
extern thread_local hipError_t g_last_hip_error;
int x6_products();     // vit_gemm_x6.hip: partial products per launch (6 / 3; 2 = "f16x3"), per host thread
void x6_take_amax(const uint32_t *&a, const uint32_t *&b);      // vit_gemm_x6.hip: the announced |max| words of the next launch (consumed)
uint32_t *x6_take_output_amax();                                 // vit_gemm_x6.hip: where the next launch publishes the |max| of its output (or null)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace x6r {
constexpr int BK = 16;

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ inline float gelu_grad_exact(float x)     // as in vit_gemm_x6.hip (act = 2: the input-gradient GEMM of the layer behind a GELU)
{
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

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

template <typename V4> __device__ inline void split8(const V4 &lo, const V4 &hi, bf16x8 &f0, bf16x8 &f1, bf16x8 &f2)
{
    uint4 q0, q1, q2;
    split2(lo.x, lo.y, q0.x, q1.x, q2.x);
    split2(lo.z, lo.w, q0.y, q1.y, q2.y);
    split2(hi.x, hi.y, q0.z, q1.z, q2.z);
    split2(hi.z, hi.w, q0.w, q1.w, q2.w);
    f0 = __builtin_bit_cast(bf16x8, q0); f1 = __builtin_bit_cast(bf16x8, q1); f2 = __builtin_bit_cast(bf16x8, q2);
}

// ---- "f16x3" (NPROD == 2; see vit_gemm_x6.hip): two fp16 pieces of value * 2^k, k from the operand tensor's |max| ---------------
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
    if (e == 0 || e == 255) return 1.f;
    const int se = min(max(127 + 14 - (e - 127), 27), 227);
    return __builtin_bit_cast(float, (uint32_t)se << 23);
}
__device__ inline void split2h(float a, float b, uint32_t &p0, uint32_t &p1)
{
    // h = RNE fp16 of the pair; l = fp16 of the exact residuals a - h (v_fma_mix_f32 reads the fp16 halves in place: no v_cvt_f32_f16)
    float ra, rb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(p0));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
}
template <int NPROD, typename V4> __device__ inline void split8s(const V4 &lo, const V4 &hi, float s, bf16x8 &f0, bf16x8 &f1, bf16x8 &f2)
{
#ifdef VIT_EXP_NOSPLIT   /* experiment builds only (tools/exp_nosplit.sh): see vit_gemm_x6.hip */
    if (NPROD == 2) { f0 = __builtin_bit_cast(bf16x8, lo); f1 = __builtin_bit_cast(bf16x8, hi); return; }
#endif
    if constexpr (NPROD == 2) {
        uint4 q0, q1;
        split2h(lo.x * s, lo.y * s, q0.x, q1.x);
        split2h(lo.z * s, lo.w * s, q0.y, q1.y);
        split2h(hi.x * s, hi.y * s, q0.z, q1.z);
        split2h(hi.z * s, hi.w * s, q0.w, q1.w);
        f0 = __builtin_bit_cast(bf16x8, q0); f1 = __builtin_bit_cast(bf16x8, q1); f2 = f1;
    } else {
        split8(lo, hi, f0, f1, f2);
    }
}
template <int NPROD> __device__ inline f32x16 mma(const bf16x8 &a, const bf16x8 &b, const f32x16 &c)
{
    if constexpr (NPROD == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// same walk as vit_gemm_x6.hip's: XCD x owns a contiguous range of the tile sequence, the sequence walks groups of 8
// row tiles column by column
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

typedef __attribute__((address_space(3))) void *lptr_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Fragment reads are inline asm: a compiler-visible ds_read behind an LDS-DMA makes hipcc wait for vmcnt(0) -- for the
// youngest DMA -- before it (it cannot tell the ring's stages apart), which would serialise the ring.  The asm reads
// return asynchronously; lds_wait<N> waits until at most N of them are outstanding and lists the registers that are valid
// from then on as in/out operands, so no consumer can be scheduled ahead of it.
template <int OFF> __device__ inline void lds_read(f32x4 &dst, uint32_t addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ inline void lds_wait(f32x4 &a, f32x4 &b)
{
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b) : [n] "n"(N) : "memory");
}
template <int N> __device__ inline void lds_wait(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e)
{
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : [n] "n"(N) : "memory");
}
template <int N> __device__ inline void lds_wait(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e, f32x4 &f, f32x4 &g, f32x4 &h)
{
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : [n] "n"(N) : "memory");
}

template <int N> __device__ inline void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x BN output tile, WM x WN wavefronts of (BM/WM) x (BN/WN) sub-tiles, NST LDS stages
template <int ACT, int BM, int BN, int WM, int WN, int NST, int OCC, int NPROD = 6>
__global__ void __launch_bounds__(64 * WM * WN, OCC) k_linear_x6r(const float *__restrict__ x, const uint4 *__restrict__ wp,
                                                                   const float *__restrict__ bias, const float *__restrict__ residual,
                                                                   float *__restrict__ out, float *__restrict__ pre, int M, int N, int K,
                                                                   const uint32_t *__restrict__ amax_x, const uint32_t *__restrict__ amax_w,
                                                                   uint32_t *__restrict__ amax_out)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the launch stub; it has no amdgcn builtins / asm constraints)
    constexpr int NW = WM * WN, RM = BM / WM / 32, RN = BN / WN / 32;
    float sx = 1.f, ix = 1.f, iw = 1.f;           // f16x3: activation scale (applied at the fragment split), inverse scales of the epilogue
    if constexpr (NPROD == 2) { sx = f16_scale(amax_line(amax_x)); ix = 1.f / sx; iw = 1.f / f16_scale(amax_line(amax_w)); }
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 96, ST_BYTES = A_BYTES + B_BYTES;
    constexpr int A_CH = BM / 16, B_CH = 6 * (BN / 64), CH = A_CH + B_CH, CPW = (CH + NW - 1) / NW;   // 1 KiB DMA chunks per stage, per wave
    static_assert(BM % 64 == 0 && BN % 64 == 0 && RM >= 1 && (RN == 1 || RN == 2), "tile shape");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int KG = K >> 3, NB = (N + 63) >> 6;
    const int nk = K / BK;

    // this wave's DMA chunks.  MUBUF (`buffer_load_dwordx4 ... lds`) rather than `global_load_lds`: the compiler keeps counted
    // vmcnt waits for buffer loads, while a pending FLAT-encoded LDS access turns every later wait into vmcnt(0).  Buffer
    // bases are the tile's first row / first 64-row weight block, so the 32-bit offsets stay far below 4 GiB for any tensor.
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (int64_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(wp + (int64_t)(n0 >> 6) * KG * 3 * 64), 0, 0x7fffffff, 0x00020000);
    int voff[CPW], dst[CPW];      // per-lane byte offset of slab 0 inside the buffer; LDS offset inside a stage (wave-uniform)
    bool is_a[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(wave + i * NW, CH - 1);                     // (a surplus chunk repeats the last one: same bytes, same place)
        is_a[i] = c < A_CH;
        if (c < A_CH) {
            const int row = c * 16 + (lane >> 2), q = (lane & 3) ^ ((row >> 2) & 3);
            voff[i] = min(row, M - 1 - m0) * K * 4 + q * 16;          // rows past M: the last row again (never stored)
            dst[i] = c * 1024;
        } else {
            const int cb = c - A_CH, nbl = cb / 6, kg = (cb % 6) / 3, p = cb % 3;
            const int nb = min(nbl, NB - 1 - (n0 >> 6));
            voff[i] = ((nb * KG + kg) * 3 + p) * 1024 + lane * 16;
            dst[i] = A_BYTES + ((kg * 3 + p) * BN + nbl * 64) * 16;
        }
    }
    auto issue = [&](int s, unsigned char *stage) {   // DMA slab min(s, nk-1) (past the end: the last slab again, into a stage nobody reads)
        const int ks = min(s, nk - 1);
        const int so_a = ks * (BK * 4), so_b = ks * (2 * 3 * 1024);
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if (is_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(stage + dst[i]), 16, voff[i], so_a, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lptr_t)(stage + dst[i]), 16, voff[i], so_b, 0, 0);
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x16{0};

    // fragment addresses inside a stage (LDS byte addresses: the fragment reads are inline asm, see lds_read)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    uint32_t a_off[RM][2];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        const int row = wm * (BM / WM) + 32 * i + col, g = (row >> 2) & 3;
        a_off[i][0] = lds0 + row * 64 + (((2 * half) ^ g) << 4);
        a_off[i][1] = lds0 + row * 64 + (((2 * half + 1) ^ g) << 4);
    }
    const uint32_t b_off = lds0 + A_BYTES + ((half * 3) * BN + wn * (BN / WN) + col) * 16;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, smem + s * ST_BYTES);

    int cur = 0;                                  // stage of slab s
    for (int s = 0; s < nk; ++s) {
        const int refill = cur == 0 ? NST - 1 : cur - 1;     // the stage slab s-1 occupied
        wait_vmcnt<CPW * (NST - 2)>();          // this wave's chunks of slab s have landed (NST-2 younger slabs stay in flight)
        asm volatile("s_barrier" ::: "memory");  // ... and everybody's; everybody is also done reading slab s-1
        issue(s + NST - 1, smem + refill * ST_BYTES);
        const uint32_t so = cur * ST_BYTES;
        f32x4 fb[RN][3], lo[2], hi[2];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) lds_read<0>(fb[j][p], b_off + so + (p * BN + 32 * j) * 16);
        lds_read<0>(lo[0], a_off[0][0] + so);
        lds_read<0>(hi[0], a_off[0][1] + so);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            if (i + 1 < RM) {
                lds_read<0>(lo[(i + 1) & 1], a_off[i + 1][0] + so);
                lds_read<0>(hi[(i + 1) & 1], a_off[i + 1][1] + so);
            }
            // everything but the two reads just issued has arrived; the operand list ties the consumers to this wait
            if (i == 0) {
                if constexpr (RN == 2) lds_wait<(RM > 1) ? 2 : 0>(fb[0][0], fb[0][1], fb[0][2], fb[1][0], fb[1][1], fb[1][2], lo[0], hi[0]);
                else lds_wait<(RM > 1) ? 2 : 0>(fb[0][0], fb[0][1], fb[0][2], lo[0], hi[0]);
            } else if (i + 1 < RM) lds_wait<2>(lo[i & 1], hi[i & 1]);
            else lds_wait<0>(lo[i & 1], hi[i & 1]);
            bf16x8 fa0, fa1, fa2;
            split8s<NPROD>(lo[i & 1], hi[i & 1], sx, fa0, fa1, fa2);
#pragma unroll
            for (int j = 0; j < RN; ++j) {      // smallest partial products first
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[j][0]), b1 = __builtin_bit_cast(bf16x8, fb[j][1]), b2 = __builtin_bit_cast(bf16x8, fb[j][2]);
                f32x16 c = acc[i][j];
                if constexpr (NPROD == 6) {     // the three 2^-16-level products; left out in three-product mode (vit_x6_set_products(3))
                    c = mma<NPROD>(fa2, b0, c);
                    c = mma<NPROD>(fa1, b1, c);
                    c = mma<NPROD>(fa0, b2, c);
                }
                c = mma<NPROD>(fa1, b0, c);
                c = mma<NPROD>(fa0, b1, c);
                c = mma<NPROD>(fa0, b0, c);
                acc[i][j] = c;
            }
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    wait_vmcnt<0>();   // the surplus DMAs of the last slabs must not outlive the workgroup's LDS allocation

    // acc[i][j]: lane column n = n0 + wn*(BN/WN) + 32 j + col ; register r = row m0 + wm*(BM/WM) + 32 i + (r&3) + 8 (r>>2) + 4 half
    uint32_t omax = 0;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (BN / WN) + 32 * j + col;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const int64_t o = (int64_t)m * N + n;
                float t = acc[i][j][r];
                if constexpr (NPROD == 2) t = t * ix * iw;
                t += bv;
                if (ACT == 2) { t *= gelu_grad_exact(residual[o]); out[o] = t; omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu); continue; }
                if (pre) pre[o] = t;
                if (ACT == 1) t = gelu_exact(t);
                if (residual) t += residual[o];
                out[o] = t;
                omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu);
            }
        }
    }
    if (amax_out) {             // |max| of the stored values (see vit_x6_set_output_amax): one atomic per wave
        amax_fold(amax_out, omax);
    }
#endif
}

template <int OFF> __device__ inline void lds_write(uint32_t addr, const bf16x8 &v)
{
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N> __device__ inline void lds_wait(f32x4 &a, f32x4 &b, f32x4 &c)
{
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c) : [n] "n"(N) : "memory");
}
template <int N> __device__ inline void lds_wait(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e, f32x4 &f, f32x4 &g, f32x4 &h, f32x4 &i)
{
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i) : [n] "n"(N) : "memory");
}

__device__ inline void lds_wait_all(f32x4 (&b)[2][3], f32x4 (&a)[4][3])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(a[0][0]), "+v"(a[0][1]),
                   "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[3][0]),
                   "+v"(a[3][1]), "+v"(a[3][2])
                 :
                 : "memory");
}

// Variant that splits every activation ONCE per workgroup: the fp32 slab arrives by DMA (raw image), each wave converts its
// 32-row share into the three bf16 planes of a second image one slab ahead of the MFMAs, and all fragment reads are plane
// reads (A like B).  Per wave and slab: 2 + 3*(RM+RN) 16-byte LDS reads, 3 writes, one split8 -- instead of RM split8's.
// LDS: raw A x2, B x2, converted A x2; every DMA is issued one whole slab before its consumer, so the top-of-slab wait is
// a plain vmcnt(0) with nothing young in flight.
template <int ACT, int BM, int BN, int WM, int WN, bool PROF = false, int NPROD = 6>
__global__ void __launch_bounds__(64 * WM * WN, 1) k_linear_x6c(const float *__restrict__ x, const uint4 *__restrict__ wp,
                                                                 const float *__restrict__ bias, const float *__restrict__ residual,
                                                                 float *__restrict__ out, float *__restrict__ pre, int M, int N, int K,
                                                                 float4 *__restrict__ slabs, int *__restrict__ tickets,
                                                                 const uint32_t *__restrict__ amax_x, const uint32_t *__restrict__ amax_w,
                                                                 uint32_t *__restrict__ amax_out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float sx = 1.f, ix = 1.f, iw = 1.f;           // f16x3: activation scale (applied by the converter), inverse scales of the epilogue
    if constexpr (NPROD == 2) { sx = f16_scale(amax_line(amax_x)); ix = 1.f / sx; iw = 1.f / f16_scale(amax_line(amax_w)); }
    constexpr int NW = WM * WN, RM = BM / WM / 32, RN = BN / WN / 32;
    constexpr int RAW_BYTES = BM * 64, B_BYTES = BN * 96, AC_BYTES = BM * 96;
    constexpr int RAW0 = 0, B0 = 3 * RAW_BYTES, AC0 = B0 + 2 * B_BYTES, LDS_BYTES = AC0 + 2 * AC_BYTES;   // raw ring of 3
    constexpr int A_CH = BM / 16, B_CH = 6 * (BN / 64), CH = A_CH + B_CH, CPW = (CH + NW - 1) / NW;
    static_assert(BM == 32 * NW, "every wave converts 32 rows of the slab");
    static_assert(RN == 2 && RM == 4, "tile shape");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int KG = K >> 3, NB = (N + 63) >> 6;
    // gridDim.y > 1: this workgroup contracts slabs [k_lo, k_lo + nk) of the K / 16 (N <= 1024 gives 84 tiles of 256 x 256 for 256
    // CUs, 84 x 3 fills them).  The S partial tiles of an output tile meet through `slabs` (register layout, 1 KiB per store
    // instruction) and a ticket counter: whoever draws the last ticket adds the other S - 1 slabs to its own accumulators and
    // runs the normal epilogue -- plain stores and loads, one release / acquire pair per tile, no atomics on the output.
    const int nk_all = K / BK, S_ = gridDim.y, sp_ = blockIdx.y;
    const int k_lo = sp_ * (nk_all / S_) + min(sp_, nk_all % S_), nk = nk_all / S_ + (sp_ < nk_all % S_ ? 1 : 0);

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (int64_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(wp + (int64_t)(n0 >> 6) * KG * 3 * 64), 0, 0x7fffffff, 0x00020000);
    int voff[CPW], dst[CPW];
    bool is_a[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = min(wave + i * NW, CH - 1);
        is_a[i] = c < A_CH;
        if (c < A_CH) {
            const int row = c * 16 + (lane >> 2), q = (lane & 3) ^ ((row >> 2) & 3);
            voff[i] = min(row, M - 1 - m0) * K * 4 + q * 16;
            dst[i] = c * 1024;
        } else {
            const int cb = c - A_CH, nbl = cb / 6, kg = (cb % 6) / 3, p = cb % 3;
            const int nb = min(nbl, NB - 1 - (n0 >> 6));
            voff[i] = ((nb * KG + kg) * 3 + p) * 1024 + lane * 16;
            dst[i] = ((kg * 3 + p) * BN + nbl * 64) * 16;
        }
    }
    // raw A slab sa -> raw[sa & 1], B slab sb -> B[sb & 1] (past the end: the last slab again, never consumed)
    auto issue = [&](int ring, int sa, int sb) {   // raw A slab sa -> raw[ring], B slab sb -> B[sb & 1]
        const int so_a = (k_lo + min(sa, nk - 1)) * (BK * 4), so_b = (k_lo + min(sb, nk - 1)) * (2 * 3 * 1024);
        unsigned char *ra = smem + RAW0 + ring * RAW_BYTES, *rb = smem + B0 + (sb & 1) * B_BYTES;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if (is_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(ra + dst[i]), 16, voff[i], so_a, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lptr_t)(rb + dst[i]), 16, voff[i], so_b, 0, 0);
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x16{0};

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // converter: lane -> (row 32 wave + col, k group half) of the raw image; its three pieces go to the planes
    const int crow = 32 * wave + col, cg = (crow >> 2) & 3;
    const uint32_t c_rd0 = lds0 + RAW0 + crow * 64 + (((2 * half) ^ cg) << 4), c_rd1 = lds0 + RAW0 + crow * 64 + (((2 * half + 1) ^ cg) << 4);
    const uint32_t c_wr = lds0 + AC0 + ((half * 3) * BM + crow) * 16;
    // fragments: planes [k group][piece][row][16 B]
    const uint32_t a_rd = lds0 + AC0 + ((half * 3) * BM + wm * (BM / WM) + col) * 16;
    const uint32_t b_rd = lds0 + B0 + ((half * 3) * BN + wn * (BN / WN) + col) * 16;

    auto convert = [&](int s_, f32x4 &lo, f32x4 &hi) {   // (lo, hi already waited for) -> planes of Ac[s_ & 1]
        bf16x8 f0, f1, f2;
        split8s<NPROD>(lo, hi, sx, f0, f1, f2);
        const uint32_t w = c_wr + (s_ & 1) * AC_BYTES;
        lds_write<0>(w, f0); lds_write<BM * 16>(w, f1);
        if constexpr (NPROD == 6) lds_write<2 * BM * 16>(w, f2);      // three-product mode never reads the third plane
    };

    // prologue: A(0), A(1), A(2), B(0) land; everybody converts its share of slab 0
    issue(0, 0, 0);
    issue(1, 1, 0);                              // (B(0) again: same bytes, same place -- keeps issue() uniform)
    issue(2, 2, 0);
    wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    f32x4 fb[RN][3], fa[RM][3], clo, chi;
    {
        f32x4 lo, hi;
        lds_read<0>(lo, c_rd0); lds_read<0>(hi, c_rd1);
        lds_read<0>(clo, c_rd0 + RAW_BYTES); lds_read<0>(chi, c_rd1 + RAW_BYTES);     // (second wave group: its share of slab 1)
        lds_wait<0>(lo, hi);
        lds_wait<0>(clo, chi);
        convert(0, lo, hi);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    long long tw = 0, ti = 0, tl = 0, tc = 0, tp = PROF ? clock64() : 0;
#define X6C_T(acc_) do { if (PROF) { const long long t_ = clock64(); acc_ += t_ - tp; tp = t_; } } while (0)
    // all of this wave's fragments of slab s into registers (+ for the first wave group: its raw share of slab s+1)
    auto load_phase = [&](int s, int raw_ring) {
        const uint32_t ao = a_rd + (s & 1) * AC_BYTES, bo = b_rd + (s & 1) * B_BYTES;
        lds_read<0>(clo, c_rd0 + raw_ring * RAW_BYTES); lds_read<0>(chi, c_rd1 + raw_ring * RAW_BYTES);
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            lds_read<0>(fb[j][0], bo + j * 512); lds_read<BN * 16>(fb[j][1], bo + j * 512); lds_read<2 * BN * 16>(fb[j][2], bo + j * 512);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            lds_read<0>(fa[i][0], ao + i * 512); lds_read<BM * 16>(fa[i][1], ao + i * 512); lds_read<2 * BM * 16>(fa[i][2], ao + i * 512);
        }
        lds_wait_all(fb, fa);
        lds_wait<0>(clo, chi);
    };
    // 48 MFMAs on the fragments in registers, with the conversion of (clo, chi) -> Ac[s_conv & 1] riding in their issue gaps:
    // a wave's own VALU work between its MFMAs is free (up to ~5 slots per 32-cycle MFMA), the same work issued by the
    // SIMD's other wave while this one streams MFMAs gets about one slot per MFMA (measured: 44 instructions = 1 600 cycles)
    auto mfma_phase = [&](int s_conv, bool do_mfma) {
        uint4 q0, q1, q2;
        uint32_t *p0 = reinterpret_cast<uint32_t *>(&q0), *p1 = reinterpret_cast<uint32_t *>(&q1), *p2 = reinterpret_cast<uint32_t *>(&q2);
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[i][0]), a1 = __builtin_bit_cast(bf16x8, fa[i][1]), a2 = __builtin_bit_cast(bf16x8, fa[i][2]);
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[j][0]), b1 = __builtin_bit_cast(bf16x8, fb[j][1]), b2 = __builtin_bit_cast(bf16x8, fb[j][2]);
                    f32x16 c = acc[i][j];
                    if constexpr (NPROD == 6) {
                        c = mma<NPROD>(a2, b0, c);
                        c = mma<NPROD>(a1, b1, c);
                        c = mma<NPROD>(a0, b2, c);
                    }
                    c = mma<NPROD>(a1, b0, c);
                    c = mma<NPROD>(a0, b1, c);
                    c = mma<NPROD>(a0, b0, c);
                    acc[i][j] = c;
                    const int blk = i * RN + j;
                    if (blk >= 4) {               // one quarter of the split rides in this block's six MFMA gaps
                        const float c0 = blk == 4 ? clo.x : blk == 5 ? clo.z : blk == 6 ? chi.x : chi.z, c1 = blk == 4 ? clo.y : blk == 5 ? clo.w : blk == 6 ? chi.y : chi.w;
                        if constexpr (NPROD == 2) split2h(c0 * sx, c1 * sx, p0[blk - 4], p1[blk - 4]);
                        else split2(c0, c1, p0[blk - 4], p1[blk - 4], p2[blk - 4]);
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            if constexpr (NPROD == 2) {
                split2h(clo.x * sx, clo.y * sx, p0[0], p1[0]); split2h(clo.z * sx, clo.w * sx, p0[1], p1[1]);
                split2h(chi.x * sx, chi.y * sx, p0[2], p1[2]); split2h(chi.z * sx, chi.w * sx, p0[3], p1[3]);
            } else {
                split2(clo.x, clo.y, p0[0], p1[0], p2[0]); split2(clo.z, clo.w, p0[1], p1[1], p2[1]);
                split2(chi.x, chi.y, p0[2], p1[2], p2[2]); split2(chi.z, chi.w, p0[3], p1[3], p2[3]);
            }
        }
        const uint32_t w = c_wr + (s_conv & 1) * AC_BYTES;
        lds_write<0>(w, __builtin_bit_cast(bf16x8, q0)); lds_write<BM * 16>(w, __builtin_bit_cast(bf16x8, q1));
        if constexpr (NPROD == 6) lds_write<2 * BM * 16>(w, __builtin_bit_cast(bf16x8, q2));
    };

    // The two waves that share a SIMD (w and w + 4) belong to the two row halves of the tile.  The first half runs
    // load -> MFMA inside a slab interval, the second half MFMA (of the previous slab, from registers) -> load: one wave's
    // matrix phase covers the other's LDS phase instead of both idling the matrix pipe at the same time.
    // ring position of raw slab s (s % 3), kept incrementally
    int r0 = 0;
    if (wave < NW / 2) {
        for (int s = 0; s < nk; ++s) {
            const int r1 = r0 == 2 ? 0 : r0 + 1;
            wait_vmcnt<0>();                          // issued a whole slab ago
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // conversions written; everybody's DMAs visible; previous reads done
            X6C_T(tw);
            issue(r0, s + 3, s + 1);
            X6C_T(ti);
            load_phase(s, r1);                        // fragments of slab s + this wave's raw share of slab s+1
            __builtin_amdgcn_sched_barrier(0);
            X6C_T(tl);
            mfma_phase(s + 1, true);                  // ... converted under the MFMAs of slab s
            __builtin_amdgcn_sched_barrier(0);
            X6C_T(tc);
            r0 = r1;
        }
    } else {
        for (int s = 0; s < nk; ++s) {
            const int r1 = r0 == 2 ? 0 : r0 + 1, r2 = r1 == 2 ? 0 : r1 + 1;
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            X6C_T(tw);
            __builtin_amdgcn_s_setprio(2);            // this group's MFMAs first; the other group is in its LDS phase
            if (s > 0) mfma_phase(s + 1, true); else mfma_phase(s + 1, false);   // MFMAs of slab s-1 (registers); converts its share of slab s+1
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            X6C_T(tc);
            issue(r0, s + 3, s + 1);                  // (half a slab after the first group's chunks: still most of a slab ahead of the wait)
            X6C_T(ti);
            load_phase(s, r2);                        // fragments of slab s + its raw share of slab s+2 (landed a slab ago)
            __builtin_amdgcn_sched_barrier(0);
            X6C_T(tl);
            r0 = r1;
        }
        mfma_phase(nk + 1, true);                 // MFMAs of the last slab (its conversion lands in a stage nobody reads any more)
    }
#undef X6C_T
    if (PROF) {
        if (pre && blockIdx.x == min(8, (int)gridDim.x - 1) && lane == 0) {      // cfg 4: `pre` receives 8 x 4 per-phase cycle counts of one workgroup
            pre[wave * 4 + 0] = (float)tw / nk; pre[wave * 4 + 1] = (float)ti / nk; pre[wave * 4 + 2] = (float)tl / nk; pre[wave * 4 + 3] = (float)tc / nk;
        }
        pre = nullptr;
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (S_ > 1) {
        const int tile_id = tm * tiles_n + tn;
        constexpr int PER_WAVE = RM * RN * 4;                 // float4 stores per lane
        auto slab_of = [&](int split) { return slabs + (((size_t)tile_id * S_ + split) * NW + wave) * (PER_WAVE * 64) + lane; };
        float4 *mine = slab_of(sp_);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    mine[((i * RN + j) * 4 + q4) * 64] = make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // (also: every wave is done with the LDS images)
        int *flag = reinterpret_cast<int *>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *flag = __hip_atomic_fetch_add(tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*flag != S_ - 1) return;                            // not the last arriver of this tile
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            tickets[tile_id] = 0;                               // ready for the next launch (the host zeroes it before the first)
        }
        __syncthreads();
        for (int t = 0; t < S_; ++t) {
            if (t == sp_) continue;
            const float4 *other = slab_of(t);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 v = other[((i * RN + j) * 4 + q4) * 64];
                        acc[i][j][4 * q4] += v.x; acc[i][j][4 * q4 + 1] += v.y; acc[i][j][4 * q4 + 2] += v.z; acc[i][j][4 * q4 + 3] += v.w;
                    }
        }
    }

    uint32_t omax = 0;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (BN / WN) + 32 * j + col;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const int64_t o = (int64_t)m * N + n;
                float t = acc[i][j][r];
                if constexpr (NPROD == 2) t = t * ix * iw;
                t += bv;
                if (ACT == 2) { t *= gelu_grad_exact(residual[o]); out[o] = t; omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu); continue; }
                if (pre) pre[o] = t;
                if (ACT == 1) t = gelu_exact(t);
                if (residual) t += residual[o];
                out[o] = t;
                omax = max(omax, __builtin_bit_cast(uint32_t, t) & 0x7fffffffu);
            }
        }
    }
    if (amax_out) {             // |max| of the stored values (vit_x6_set_output_amax): one atomic per wave
        amax_fold(amax_out, omax);
    }
#endif
}

// w (R, C) row-major fp32 -> block layout packed[r / 64][c / 8][piece][r % 64][8] bf16, rows padded with zeros to a
// multiple of 64.  transpose = 1 packs w^T (output row = column of w, k = row of w).
template <int NPROD>
__global__ void __launch_bounds__(256) k_split_block(const float *__restrict__ w, uint4 *__restrict__ packed, int rows, int cols,
                                                     int transpose, const uint32_t *__restrict__ amax, uint32_t *__restrict__ tail)
{
    if (NPROD == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) tail[threadIdx.x * AMAX_STRIDE] = amax[threadIdx.x * AMAX_STRIDE];     // the line the readers of this image take their inverse scale from
    const float sw = NPROD == 2 ? f16_scale(amax_line(amax)) : 1.f;
    // output rows R_ = transpose ? cols : rows, contraction length K_ = transpose ? rows : cols
    const int R_ = transpose ? cols : rows, K_ = transpose ? rows : cols, KG = K_ >> 3;
    __shared__ float s[64][65];                       // [output row][k]: 64 rows x 64 k (8 k groups)
    const int rb = blockIdx.y, k0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int r, k;
        if (transpose) { k = i >> 6; r = i & 63; } else { r = i >> 6; k = i & 63; }      // coalesced along the source's fast axis
        const int R = rb * 64 + r, Kx = k0 + k;
        float v = 0.f;
        if (R < R_ && Kx < K_) v = transpose ? w[(int64_t)Kx * cols + R] : w[(int64_t)R * cols + Kx];
        s[r][k] = v;
    }
    __syncthreads();
    // thread -> (k group of the 64-wide slice, output row): consecutive threads = consecutive rows of one plane
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        const int kg = i >> 6, r = i & 63;
        if (k0 + kg * 8 >= K_) continue;
        const float4 lo = make_float4(s[r][kg * 8 + 0], s[r][kg * 8 + 1], s[r][kg * 8 + 2], s[r][kg * 8 + 3]);
        const float4 hi = make_float4(s[r][kg * 8 + 4], s[r][kg * 8 + 5], s[r][kg * 8 + 6], s[r][kg * 8 + 7]);
        bf16x8 f0, f1, f2;
        split8s<NPROD>(lo, hi, sw, f0, f1, f2);
        uint4 *o = packed + (((int64_t)rb * KG + (k0 >> 3) + kg) * 3) * 64 + r;
        o[0] = __builtin_bit_cast(uint4, f0); o[64] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[128] = __builtin_bit_cast(uint4, f2);
    }
}
// BOTH images of one Linear weight in one launch (round 6): the forward image (output rows = rows of w, contraction along its columns) and
// the transposed one for the input-gradient GEMM (output rows = columns of w), each in the block layout above (flag set) or in the
// MFMA-order row layout of vit_split_weight (packed[row][k / 8][piece][8]).  After an optimizer step every trainable weight needs both
// again; one launch per image read the fp32 weight twice and was ~1 270 launches / 12 ms of a 281 ms C3 step, launch-bound.  A workgroup owns
// a 64 x 64 tile of w: it IS a 64-row x 64-k tile of the forward image and a 64-row x 64-k tile of the transposed one.  Same split
// function on the same values: the bytes equal the single-image kernels' (tests/test_gpu_vit.py compares them).
template <int NPROD>
__global__ void __launch_bounds__(256) k_split_pair(const float *__restrict__ w, uint4 *__restrict__ pf, uint4 *__restrict__ pt, int rows, int cols,
                                                    int block_f, int block_t, const uint32_t *__restrict__ amax, uint32_t *__restrict__ tail_f,
                                                    uint32_t *__restrict__ tail_t)
{
    if (NPROD == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        const uint32_t a = amax[threadIdx.x * AMAX_STRIDE];
        tail_f[threadIdx.x * AMAX_STRIDE] = a; tail_t[threadIdx.x * AMAX_STRIDE] = a;
    }
    const float sw = NPROD == 2 ? f16_scale(amax_line(amax)) : 1.f;
    __shared__ float s[64][65];                       // [row of w][column of w]
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    if ((cols & 3) == 0) {                            // four columns per load (rows of w are 16-byte aligned)
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, c4 = (i & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < rows && c0 + c4 < cols) v = *reinterpret_cast<const float4 *>(w + (int64_t)(r0 + r) * cols + c0 + c4);
            s[r][c4] = v.x; s[r][c4 + 1] = v.y; s[r][c4 + 2] = v.z; s[r][c4 + 3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            s[r][c] = (r0 + r < rows && c0 + c < cols) ? w[(int64_t)(r0 + r) * cols + c0 + c] : 0.f;
        }
    }
    __syncthreads();
    // ---- forward image: output row = row of w, k = column of w ----
    {
        const int KG = cols >> 3;
        for (int i = threadIdx.x; i < 8 * 64; i += 256) {
            const int kg = block_f ? (i >> 6) : (i & 7), r = block_f ? (i & 63) : (i >> 3);
            if (c0 + kg * 8 >= cols) continue;
            const float4 lo = make_float4(s[r][kg * 8 + 0], s[r][kg * 8 + 1], s[r][kg * 8 + 2], s[r][kg * 8 + 3]);
            const float4 hi = make_float4(s[r][kg * 8 + 4], s[r][kg * 8 + 5], s[r][kg * 8 + 6], s[r][kg * 8 + 7]);
            bf16x8 f0, f1, f2;
            split8s<NPROD>(lo, hi, sw, f0, f1, f2);
            if (block_f) {
                uint4 *o = pf + (((int64_t)blockIdx.y * KG + (c0 >> 3) + kg) * 3) * 64 + r;
                o[0] = __builtin_bit_cast(uint4, f0); o[64] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[128] = __builtin_bit_cast(uint4, f2);
            } else if (r0 + r < rows) {
                uint4 *o = pf + ((int64_t)(r0 + r) * KG + (c0 >> 3) + kg) * 3;
                o[0] = __builtin_bit_cast(uint4, f0); o[1] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[2] = __builtin_bit_cast(uint4, f2);
            }
        }
    }
    // ---- transposed image: output row = column of w, k = row of w ----
    {
        const int KG = rows >> 3;
        for (int i = threadIdx.x; i < 8 * 64; i += 256) {
            const int rg = block_t ? (i >> 6) : (i & 7), c = block_t ? (i & 63) : (i >> 3);
            if (r0 + rg * 8 >= rows) continue;
            const float4 lo = make_float4(s[rg * 8 + 0][c], s[rg * 8 + 1][c], s[rg * 8 + 2][c], s[rg * 8 + 3][c]);
            const float4 hi = make_float4(s[rg * 8 + 4][c], s[rg * 8 + 5][c], s[rg * 8 + 6][c], s[rg * 8 + 7][c]);
            bf16x8 f0, f1, f2;
            split8s<NPROD>(lo, hi, sw, f0, f1, f2);
            if (block_t) {
                uint4 *o = pt + (((int64_t)blockIdx.x * KG + (r0 >> 3) + rg) * 3) * 64 + c;
                o[0] = __builtin_bit_cast(uint4, f0); o[64] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[128] = __builtin_bit_cast(uint4, f2);
            } else if (c0 + c < cols) {
                uint4 *o = pt + ((int64_t)(c0 + c) * KG + (r0 >> 3) + rg) * 3;
                o[0] = __builtin_bit_cast(uint4, f0); o[1] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[2] = __builtin_bit_cast(uint4, f2);
            }
        }
    }
}

// All the weight images of a model in ONE launch (after an optimizer step every weight changed: the per-weight launches above were
// 1 274 launches and 12 ms of a 285 ms train step at 1.4 TB/s -- launch-bound).  A job = one image of one weight; a workgroup = one
// 64-row x 64-k tile of one job (binary search over the jobs' first-block table), staged through LDS exactly as k_split_block does, then
// written in the job's layout: kind bit 1 = the BLOCK layout above, else the MFMA-order layout of vit_split_weight
// (packed[row][k / 8][piece][8]); kind bit 0 = pack w^T.  Same bytes as the single-weight kernels (tests/test_gpu_vit.py compares them).
template <int NPROD>
__global__ void __launch_bounds__(256) k_split_many(const VitSplitJob *__restrict__ jobs, int njobs)
{
    int lo = 0, hi = njobs - 1;                        // last job whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const VitSplitJob jb = jobs[lo];
    const uint32_t local = blockIdx.x - jb.first_block;
    const int bx = (int)(local % jb.nbx), rb = (int)(local / jb.nbx);
    const int rows = jb.rows, cols = jb.cols, transpose = jb.kind & 1, block = jb.kind & 2;
    const float *__restrict__ w = jb.w;
    uint4 *__restrict__ packed = static_cast<uint4 *>(jb.packed);
    if (NPROD == 2 && local == 0 && threadIdx.x < 64) jb.tail[threadIdx.x * AMAX_STRIDE] = jb.amax[threadIdx.x * AMAX_STRIDE];
    const float sw = NPROD == 2 ? f16_scale(amax_line(jb.amax)) : 1.f;
    const int R_ = transpose ? cols : rows, K_ = transpose ? rows : cols, KG = K_ >> 3;
    __shared__ float s[64][65];
    const int k0 = bx * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int r, k;
        if (transpose) { k = i >> 6; r = i & 63; } else { r = i >> 6; k = i & 63; }
        const int R = rb * 64 + r, Kx = k0 + k;
        float v = 0.f;
        if (R < R_ && Kx < K_) v = transpose ? w[(int64_t)Kx * cols + R] : w[(int64_t)R * cols + Kx];
        s[r][k] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        // block layout: consecutive threads = consecutive rows of one (k group, piece) plane; row-major layout: consecutive threads =
        // consecutive k groups of one row (48 contiguous bytes each)
        const int kg = block ? (i >> 6) : (i & 7), r = block ? (i & 63) : (i >> 3);
        if (k0 + kg * 8 >= K_) continue;
        const float4 lo4 = make_float4(s[r][kg * 8 + 0], s[r][kg * 8 + 1], s[r][kg * 8 + 2], s[r][kg * 8 + 3]);
        const float4 hi4 = make_float4(s[r][kg * 8 + 4], s[r][kg * 8 + 5], s[r][kg * 8 + 6], s[r][kg * 8 + 7]);
        bf16x8 f0, f1, f2;
        split8s<NPROD>(lo4, hi4, sw, f0, f1, f2);
        if (block) {
            uint4 *o = packed + (((int64_t)rb * KG + (k0 >> 3) + kg) * 3) * 64 + r;
            o[0] = __builtin_bit_cast(uint4, f0); o[64] = __builtin_bit_cast(uint4, f1); if (NPROD != 2) o[128] = __builtin_bit_cast(uint4, f2);
        } else if (rb * 64 + r < R_) {
            uint4 *o = packed + ((int64_t)(rb * 64 + r) * KG + (k0 >> 3) + kg) * 3;
            o[0] = __builtin_bit_cast(uint4, f0); o[1] = __builtin_bit_cast(uint4, f1);
            if (NPROD != 2) o[2] = __builtin_bit_cast(uint4, f2);      // (f16x3 never reads slot 2)
        }
    }
}
}  // namespace x6r

int split_weights_many(const VitSplitJob *jobs_dev, int njobs, uint32_t total_blocks, hipStream_t stream)
{
    if (!jobs_dev || njobs <= 0 || total_blocks == 0) return VIT_EINVAL;
    (void)hipGetLastError();
    const int np = x6_products();
    if (np == 2) hipLaunchKernelGGL(x6r::k_split_many<2>, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, njobs);
    else if (np == 3) hipLaunchKernelGGL(x6r::k_split_many<3>, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, njobs);
    else hipLaunchKernelGGL(x6r::k_split_many<6>, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, njobs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

// both images of a Linear weight (rows x cols) in one launch: see k_split_pair.  block_f / block_t: the block layout (1) or the row layout (0)
// of the forward / the transposed image; sizes as vit_split_weight_block_bytes / vit_split_weight_bytes say.
int split_weight_pair(const float *w, void *packed_f, void *packed_t, int rows, int cols, int block_f, int block_t, hipStream_t stream)
{
    if (!w || !packed_f || !packed_t || rows <= 0 || cols <= 0 || (rows % 8) != 0 || (cols % 8) != 0) return VIT_EINVAL;
    (void)hipGetLastError();
    const uint32_t *am, *unused;
    x6_take_amax(am, unused);
    auto tail_of = [](void *p, int R_, int K_, int block) {
        const size_t pieces = block ? (size_t)((R_ + 63) / 64) * 64 * (size_t)K_ * 6 : (size_t)R_ * (size_t)K_ * 6;
        return reinterpret_cast<uint32_t *>(static_cast<char *>(p) + pieces);
    };
    uint32_t *tf = tail_of(packed_f, rows, cols, block_f), *tt = tail_of(packed_t, cols, rows, block_t);
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    const int np = x6_products();
    if (np == 2) {
        if (!am) return VIT_EINVAL;        // f16x3: the weight's |max| word must be announced (vit_x6_set_operand_amax(word, NULL))
        hipLaunchKernelGGL(x6r::k_split_pair<2>, grid, dim3(256), 0, stream, w, static_cast<uint4 *>(packed_f), static_cast<uint4 *>(packed_t), rows, cols, block_f, block_t, am, tf, tt);
    } else
        hipLaunchKernelGGL(x6r::k_split_pair<6>, grid, dim3(256), 0, stream, w, static_cast<uint4 *>(packed_f), static_cast<uint4 *>(packed_t), rows, cols, block_f, block_t, am, tf, tt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

int split_weight_block(const float *w, void *packed, int rows, int cols, int transpose, hipStream_t stream)
{
    if (!w || !packed || rows <= 0 || cols <= 0) return VIT_EINVAL;
    const int R_ = transpose ? cols : rows, K_ = transpose ? rows : cols;
    if (K_ % 8 != 0) return VIT_EINVAL;
    (void)hipGetLastError();
    // f16x3: the weight's |max| word must be announced (vit_x6_set_operand_amax(word, NULL)); the image keeps a copy right behind its pieces
    const uint32_t *am, *unused;
    x6_take_amax(am, unused);
    uint32_t *tail = reinterpret_cast<uint32_t *>(static_cast<char *>(packed) + (size_t)((R_ + 63) / 64) * 64 * (size_t)K_ * 6);
    if (x6_products() == 2) {
        if (!am) return VIT_EINVAL;
        hipLaunchKernelGGL(x6r::k_split_block<2>, dim3((K_ + 63) / 64, (R_ + 63) / 64), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), rows, cols, transpose, am, tail);
    } else
        hipLaunchKernelGGL(x6r::k_split_block<6>, dim3((K_ + 63) / 64, (R_ + 63) / 64), dim3(256), 0, stream, w, static_cast<uint4 *>(packed), rows, cols, transpose, am, tail);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

// cfg: 1 = ring kernel, 128 x 128 tiles, 3 stages, two workgroups per CU; 2 = ring kernel, 256 x 256 tiles (8 waves);
// 3 = convert-once ping-pong kernel, 256 x 256 tiles.  (tools/probes/gemm_lab.py sweeps them against vit_linear_x6_fwd.)
// vit_gemm_sm.hip: the small-M kernel on the block image (cfg 5: no split-K, no zero fill, fused epilogue, |max| word)
int linear_sm_fwd(const float *x, const void *wpb, const float *bias, const float *residual, float *out, float *pre, int M, int N, int K, int act,
                  const uint32_t *am_x, const uint32_t *am_w, uint32_t *am_out, hipStream_t stream);

int linear_x6r_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N,
                   int K, int act, int cfg, hipStream_t stream)
{
    if (!x || !wp || !out) return VIT_EINVAL;
    if (cfg == 5) {
        if (M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2 || (act == 2 && (!residual || pre))) return VIT_EINVAL;
        const uint32_t *ax, *unused;
        x6_take_amax(ax, unused);
        const uint32_t *aw = reinterpret_cast<const uint32_t *>(static_cast<const char *>(wp) + (size_t)((N + 63) / 64) * 64 * (size_t)K * 6);
        return linear_sm_fwd(x, wp, bias, residual, out, pre, M, N, K, act, ax, aw, x6_take_output_amax(), stream);
    }
    if (M <= 0 || N <= 0 || K <= 0 || (K % x6r::BK) != 0 || act < 0 || act > 2 || (act == 2 && (!residual || pre || (cfg != 1 && cfg != 3))) ||
        !((cfg >= 1 && cfg <= 4) || (cfg >= 34 && cfg <= 40))) return VIT_EINVAL;   // 32 + S: cfg 3 with an S-way K split; act 2: see vit_linear_x6_fwd
    const uint4 *w4 = static_cast<const uint4 *>(wp);
    const uint32_t *am_x, *am_unused;
    x6_take_amax(am_x, am_unused);
    const uint32_t *am_w = reinterpret_cast<const uint32_t *>(static_cast<const char *>(wp) + (size_t)((N + 63) / 64) * 64 * (size_t)K * 6);
    const bool f16 = x6_products() == 2;
    uint32_t *am_out = x6_take_output_amax();
    if (am_out && cfg != 1 && cfg != 3) return VIT_EINVAL;   // the lockstep 256 x 256 kernel (cfg 2) does not publish its output's |max|
    if (f16 && ((cfg != 1 && cfg != 3) || !am_x)) return VIT_EINVAL;      // f16x3: cfg 1 / cfg 3, with the activation's |max| announced
    (void)hipGetLastError();
#define X6R_ARGS(BM, BN, THREADS) dim3(((M + BM - 1) / BM) * ((N + BN - 1) / BN)), dim3(THREADS), 0, stream, x, w4, bias, residual, out, pre, M, N, K
    const bool three = x6_products() == 3;
    if (cfg == 1) {
        if (f16) {
            if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6r<2, 128, 128, 2, 2, 3, 2, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
            else if (act) hipLaunchKernelGGL((x6r::k_linear_x6r<1, 128, 128, 2, 2, 3, 2, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
            else hipLaunchKernelGGL((x6r::k_linear_x6r<0, 128, 128, 2, 2, 3, 2, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
        } else if (three) {
            if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6r<2, 128, 128, 2, 2, 3, 2, 3>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
            else if (act) hipLaunchKernelGGL((x6r::k_linear_x6r<1, 128, 128, 2, 2, 3, 2, 3>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
            else hipLaunchKernelGGL((x6r::k_linear_x6r<0, 128, 128, 2, 2, 3, 2, 3>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
        } else if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6r<2, 128, 128, 2, 2, 3, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
        else if (act) hipLaunchKernelGGL((x6r::k_linear_x6r<1, 128, 128, 2, 2, 3, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
        else hipLaunchKernelGGL((x6r::k_linear_x6r<0, 128, 128, 2, 2, 3, 2>), X6R_ARGS(128, 128, 256), am_x, am_w, am_out);
    } else if (cfg == 2) {
        if (act == 2) return VIT_EINVAL;
        if (act) hipLaunchKernelGGL((x6r::k_linear_x6r<1, 256, 256, 2, 4, 3, 1>), X6R_ARGS(256, 256, 512), am_x, am_w, am_out);
        else hipLaunchKernelGGL((x6r::k_linear_x6r<0, 256, 256, 2, 4, 3, 1>), X6R_ARGS(256, 256, 512), am_x, am_w, am_out);
    } else if (cfg == 4) {      // phase-timing instantiation (tools/probes/gemm_lab.py): needs `pre` with room for 32 floats
        if (!pre || (size_t)M * N < 32) return VIT_EINVAL;
        hipLaunchKernelGGL((x6r::k_linear_x6c<0, 256, 256, 2, 4, true>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
    } else if (cfg >= 34) {
        return VIT_EINVAL;            // K splits need a workspace: vit_linear_x6c_fwd
    } else if (f16) {
        if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6c<2, 256, 256, 2, 4, false, 2>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else if (act) hipLaunchKernelGGL((x6r::k_linear_x6c<1, 256, 256, 2, 4, false, 2>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else hipLaunchKernelGGL((x6r::k_linear_x6c<0, 256, 256, 2, 4, false, 2>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
    } else if (three) {
        if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6c<2, 256, 256, 2, 4, false, 3>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else if (act) hipLaunchKernelGGL((x6r::k_linear_x6c<1, 256, 256, 2, 4, false, 3>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else hipLaunchKernelGGL((x6r::k_linear_x6c<0, 256, 256, 2, 4, false, 3>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
    } else {
        if (act == 2) hipLaunchKernelGGL((x6r::k_linear_x6c<2, 256, 256, 2, 4>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else if (act) hipLaunchKernelGGL((x6r::k_linear_x6c<1, 256, 256, 2, 4>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
        else hipLaunchKernelGGL((x6r::k_linear_x6c<0, 256, 256, 2, 4>), X6R_ARGS(256, 256, 512), nullptr, nullptr, am_x, am_w, am_out);
    }
#undef X6R_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}

// The ping-pong kernel with an S-way K split whose partial tiles meet in `workspace` (x6c_workspace_bytes): S = 1 needs none.
size_t x6c_workspace_bytes(int M, int N, int splits)
{
    const size_t tiles = (size_t)((M + 255) / 256) * ((N + 255) / 256);
    return splits <= 1 ? 0 : tiles * splits * 256 * 256 * sizeof(float) + tiles * sizeof(int);
}

// Number of K splits that fills the 256 CUs best with 256 x 256 tiles (one workgroup per CU), or 0 when no split count gets
// within 20 % of whole rounds -- the caller then keeps the 128-wide kernel.  At least 16 slabs per split.
int x6c_choose_splits(int M, int N, int K)
{
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), nk = K / x6r::BK;
    int best = 0;
    float best_score = 0.8f;
    for (int S = 1; S <= 8; ++S) {
        if (S > 1 && nk / S < 16) break;
        const int wg = tiles * S, rounds = (wg + 255) / 256;
        const float score = (float)wg / (float)(rounds * 256) - 0.02f * (float)(S - 1);
        if (score > best_score) { best_score = score; best = S; }
    }
    return best;
}

int linear_x6c_fwd(const float *x, const void *wp, const float *bias, const float *residual, float *out, float *pre, int M, int N, int K,
                   int act, int splits, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (!x || !wp || !out) return VIT_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || (K % x6r::BK) != 0 || act < 0 || act > 1 || splits < 1 || splits > 8 || K / x6r::BK < splits) return VIT_EINVAL;
    if (splits > 1 && (!workspace || workspace_bytes < x6c_workspace_bytes(M, N, splits))) return VIT_EINVAL;
    if (x6_products() == 2) return VIT_EINVAL;
    const uint4 *w4 = static_cast<const uint4 *>(wp);
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    float4 *slabs = static_cast<float4 *>(workspace);
    int *tickets = splits > 1 ? reinterpret_cast<int *>(static_cast<unsigned char *>(workspace) + (size_t)tiles * splits * 256 * 256 * sizeof(float)) : nullptr;
    (void)hipGetLastError();
    if (splits > 1 && hipMemsetAsync(tickets, 0, (size_t)tiles * sizeof(int), stream) != hipSuccess) { g_last_hip_error = hipGetLastError(); return VIT_ELAUNCH; }
    if (act) hipLaunchKernelGGL((x6r::k_linear_x6c<1, 256, 256, 2, 4>), dim3(tiles, splits), dim3(512), 0, stream, x, w4, bias, residual, out, pre, M, N, K, slabs, tickets, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((x6r::k_linear_x6c<0, 256, 256, 2, 4>), dim3(tiles, splits), dim3(512), 0, stream, x, w4, bias, residual, out, pre, M, N, K, slabs, tickets, nullptr, nullptr, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = e; return VIT_ELAUNCH; }
    return VIT_OK;
}
}  // namespace vit
