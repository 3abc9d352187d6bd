// permlane_probe.hip -- what gfx950's v_permlane32_swap / v_permlane16_swap return through the clang builtins, lane by lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned u = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(u, u + 100, false, false);
    const auto d = __builtin_amdgcn_permlane16_swap(u, u + 100, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
    o[256 + threadIdx.x] = c[0]; o[320 + threadIdx.x] = c[1]; o[384 + threadIdx.x] = d[0]; o[448 + threadIdx.x] = d[1];
}
int main() {
    unsigned *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"p32(u,u)[0]", "p32(u,u)[1]", "p16(u,u)[0]", "p16(u,u)[1]", "p32(u,u+100)[0]", "p32(u,u+100)[1]", "p16(u,u+100)[0]", "p16(u,u+100)[1]"};
    for (int r = 0; r < 8; r++) { printf("%-16s", names[r]); for (int l = 0; l < 64; l += 1) printf(" %u", h[64 * r + l]); printf("\n"); }
    return 0;
}
