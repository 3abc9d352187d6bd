#include "../../sam-decoding_amd/csrc/prefill_attn_device.h"
#include <cstdio>
__global__ void k(float *o) {
    const int l = threadIdx.x;
    const float x = (l == 0) ? 1.f : ((l & 15) == 3 ? (float)(l >> 4) + 1.f : 0.f);
    o[l] = prefillattn::row4_sum(x);
    o[64 + l] = prefillattn::row4_max(x);
    float two[2] = {x, x * 2.f};
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < 2; f++) acc += prefillattn::row4_sum(two[f]) * (f + 1);
    o[128 + l] = acc;
}
int main() {
    float *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; r++) { printf("%s", r == 0 ? "sum " : r == 1 ? "max " : "mix "); for (int l = 0; l < 64; l++) printf(" %g", h[64 * r + l]); printf("\n"); }
    return 0;
}
