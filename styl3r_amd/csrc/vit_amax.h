// vit_amax.h -- the |max| "word" of the f16x3 arithmetic (vit_gemm_x6.hip has the full story): 64 uint32 slots, ONE PER 128-BYTE CACHE LINE (8 KiB),
// producers fold the bit pattern of their largest |value| into slot (workgroup + wave) & 63 with one guarded atomicMax per wave, readers take
// the max over the slots.  Shared by the kernels outside vit_gemm_x6*.hip that publish the |max| of what they store (attention, round 6).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vit {
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
