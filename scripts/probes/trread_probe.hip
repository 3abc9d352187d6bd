// trread_probe.hip -- what gfx950's ds_read_b64_tr_b16 returns (through __builtin_amdgcn_ds_read_tr16_b64_v4i16): a row-major [16 rows][64 halfs]
// tile holding value = row * 64 + col; lane L of a 16-lane group gives the address of row 4 g + L / 4, columns 4 (L % 4) .. + 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(float *o) {
    __shared__ _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int row = 4 * (l >> 4) + ((l & 15) >> 2), col = 4 * (l & 3);
    auto *p = (__attribute__((address_space(3))) s4 *)(&lds[row * 64 + col]);
    const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; j++) o[l * 4 + j] = (float)__builtin_bit_cast(_Float16, v[j]);
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) { printf("lane %2d (addr row %2d col %2d):", l, 4 * (l >> 4) + ((l & 15) >> 2), 4 * (l & 3)); for (int j = 0; j < 4; j++) printf(" (r%d,c%d)", (int)h[l * 4 + j] / 64, (int)h[l * 4 + j] % 64); printf("\n"); }
    return 0;
}
