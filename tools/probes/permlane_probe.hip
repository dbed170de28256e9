// probe: semantics of permlane32_swap / permlane16_swap and wave_reduce10 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../styl3r_amd/csrc/gsr_common.h"
__global__ void probe(unsigned *out)
{
    unsigned lane = threadIdx.x;
    unsigned x = 100 + lane, y = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[128 + lane] = s[0]; out[192 + lane] = s[1];
    float a[10], t[3];
    for (int i = 0; i < 10; ++i) a[i] = (float)((i + 1) * 1000 + lane);
    gsr::wave_reduce10(a, t);
    float *f = (float *)(out + 256);
    f[lane * 3 + 0] = t[0]; f[lane * 3 + 1] = t[1]; f[lane * 3 + 2] = t[2];
    out[256 + 192 + lane] = (unsigned)gsr::reduce10_slot(lane);
}
int main()
{
    unsigned *d; hipMalloc(&d, 4 * 600);
    probe<<<1, 64>>>(d);
    unsigned h[600]; hipMemcpy(h, d, 4 * 600, hipMemcpyDeviceToHost);
    const char *names[4] = {"p32.r0", "p32.r1", "p16.r0", "p16.r1"};
    for (int k = 0; k < 4; ++k) { printf("%s:", names[k]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * k + l]); printf("\n"); }
    float *f = (float *)(h + 256);
    for (int l = 0; l < 64; l += 16) for (int s = 0; s < 3; ++s)
        printf("lane %d tot[%d]=%.0f slot=%d\n", l + s, s, f[(l + s) * 3 + s], (int)h[256 + 192 + l + s]);
    for (int i = 0; i < 10; ++i) printf("expect a%d sum = %.0f\n", i, (double)((i + 1) * 1000 * 64 + 2016));
    return 0;
}
