// vit_amax.h -- the |max| "word" of the f16x3 arithmetic (vit_gemm_x6.hip has the full story): 64 uint32 slots, ONE PER 128-BYTE CACHE LINE (8 KiB),
// producers fold the bit pattern of their largest |value| into slot (workgroup + wave) & 63 with one guarded atomicMax per wave, readers take
// the max over the slots.  Shared by the kernels outside vit_gemm_x6*.hip that publish the |max| of what they store (attention, round 6).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vit {
// Twenty wait states.  hipcc (ROCm 7.2) inserts the "XDL write VGPR -> VALU read" wait states (11 for an 8-pass MFMA) only along the
// fall-through path of a block: where a wave-uniform branch skips a second MFMA chain and joins in front of VALU code that reads the FIRST
// chain's accumulator, the listing shows `s_nop 3` and the last accumulator registers are read before the matrix pipe has written them
// (round 6: the f16x3 attention forward un-scales S right behind its MFMAs and got keys 26, 27, 30, 31 of a tile wrong whenever the tile's
// second 32-key block was skipped; tools/probes/attn_f16_debug2.py).  Placed by hand where an accumulator is read across such a join / a loop exit.
__device__ inline void mfma_result_fence() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
constexpr int AMAX_WORD_STRIDE = 32;        // uint32 words between the 64 slots
__device__ inline uint32_t abs_bits(float x) { return __builtin_bit_cast(uint32_t, x) & 0x7fffffffu; }
// m: this lane's maximum (bit pattern of |x|); all 64 lanes of the wave call it together
__device__ inline void amax_word_fold(uint32_t *__restrict__ word, uint32_t m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    uint32_t *w = word + ((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z + (threadIdx.x >> 6)) & 63u) * AMAX_WORD_STRIDE;
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, m);
}
}  // namespace vit

// ---- "f16x3" operand arithmetic outside the GEMM files (attention, round 6): two fp16 pieces of value * s, s = the power of two that puts the
// operand tensor's |max| into [2^14, 2^15) (vit_gemm_x6.hip: f16_scale / split2h -- the same functions, so a tensor split here and there gets the
// same pieces) ----
namespace vit {
__device__ inline uint32_t amax_word_read(const uint32_t *__restrict__ word)        // all 64 lanes: the max over the 64 slots
{
    uint32_t m = word[(threadIdx.x & 63) * AMAX_WORD_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    return m;
}
__device__ inline float f16_scale_of(uint32_t amax_bits)
{
    const int e = (int)((amax_bits >> 23) & 0xff);
    if (e == 0 || e == 255) return 1.f;        // all-zero / denormal tensor; Inf / NaN inside (those propagate on their own)
    const int se = min(max(127 + 14 - (e - 127), 27), 227);       // s in [2^-100, 2^100]
    return __builtin_bit_cast(float, (uint32_t)se << 23);
}
// two (scaled) fp32 values -> their two fp16 pieces, each packed (low half = first value): h = RNE fp16, l = fp16 of the exact residual
// (a - h is exact in fp32).  Plain vector conversions, NOT the inline-asm form of vit_gemm_x6.hip (same bits): here the pieces feed an MFMA
// straight from registers, and the compiler only inserts the VALU-write -> MFMA-read wait states for instructions it can see -- with the asm
// form the forward's last key tile read a stale B operand whenever its second 32-key block was skipped (round 6, tools/probes/attn_f16_debug.py).
__device__ inline void f16_split2(float a, float b, uint32_t &p0, uint32_t &p1)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    const h2 h = __builtin_convertvector(f, h2);
    const f2 r = f - __builtin_convertvector(h, f2);
    const h2 l = __builtin_convertvector(r, h2);
    p0 = __builtin_bit_cast(uint32_t, h); p1 = __builtin_bit_cast(uint32_t, l);
}
}  // namespace vit
