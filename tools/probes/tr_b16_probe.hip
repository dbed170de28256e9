// What ds_read_b64_tr_b16 returns (gfx950): LDS holds u16 value = its own element index; lane l passes the byte address 8 l
// (its "own" four contiguous elements 4 l .. 4 l + 3); print the four elements each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out, int stride_bytes)
{
    __shared__ uint16_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)s + threadIdx.x * stride_bytes;   // LDS byte address
    uint32_t lo, hi;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
    out[threadIdx.x * 4 + 0] = lo & 0xffff; out[threadIdx.x * 4 + 1] = lo >> 16;
    out[threadIdx.x * 4 + 2] = hi & 0xffff; out[threadIdx.x * 4 + 3] = hi >> 16;
}
int main()
{
    uint32_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int stride : {8, 32}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("per-lane address stride %d bytes (lane l owns elements %d l .. +3)\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   |   ");
    }
    return 0;
}
