// trread_probe.hip -- what gfx950's ds_read_b64_tr_b16 returns (through __builtin_amdgcn_ds_read_tr16_b64_v4i16): a row-major [16 rows][64 halfs]
// tile holding value = row * 64 + col; lane L of a 16-lane group gives the address of row 4 g + L / 4, columns 4 (L % 4) .. + 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(float *o, int pattern) {
    __shared__ _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)i;
    __syncthreads();
    const int l = threadIdx.x;
    int row = 4 * (l >> 4) + ((l & 15) >> 2), col = 4 * (l & 3);
    if (pattern == 1) { row = l & 15; col = 4 * (l >> 4); }             // every lane of a group its own row, one 4-wide column block per group
    auto *p = (__attribute__((address_space(3))) s4 *)(&lds[row * 64 + col]);
    const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; j++) o[l * 4 + j] = (float)__builtin_bit_cast(_Float16, v[j]);
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int pattern = 0; pattern < 2; pattern++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pattern);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pattern);
        for (int l = 0; l < 32; l++) { printf("lane %2d:", l); for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", (int)h[l * 4 + j] / 64, (int)h[l * 4 + j] % 64); printf("\n"); }
    }
    return 0;
}
