// xcc_probe.hip -- which XCD does workgroup `lin` of a launch run on, as a function of the launches before it on the stream?
// A sequence of kernels with the grid sizes of one decoder layer (not all multiples of 8) is replayed; every workgroup records its
// hardware XCC id.  Printed per launch: the offset o such that xcc == (lin + o) % 8 for all workgroups (or "irregular").
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/xcc_probe scripts/probes/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_rec(int *out, int spin) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = x & 15;
    for (int i = 0; i < spin; i++) asm volatile("s_sleep 8");
}
int main() {
    struct L { int gx, gy, threads; const char *name; };
    const L seq[] = {{208, 1, 512, "rmsnorm+warm"}, {96, 2, 512, "qkv gemm"}, {16, 96, 64, "rope_kv"}, {32, 16, 256, "tree_attention"},
                     {16, 48, 128, "combine+warm"}, {32, 8, 512, "o gemm"}, {188, 1, 512, "rmsnorm+warm"}, {172, 1, 512, "gate|up gemm"}, {32, 8, 512, "down gemm"}};
    const int n = sizeof(seq) / sizeof(seq[0]);
    int *d; hipMalloc(&d, n * 4096 * 4);
    std::vector<int> h(n * 4096);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        if (mode == 1) hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int rep = 0; rep < (mode == 1 ? 1 : 2); rep++)
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_rec, dim3(seq[i].gx, seq[i].gy), dim3(seq[i].threads), 0, st, d + i * 4096, 20);
        if (mode == 1) { hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); hipGraphLaunch(ge, st); hipGraphLaunch(ge, st); }
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, n * 4096 * 4, hipMemcpyDeviceToHost);
        printf("%s\n", mode ? "hipGraph replay:" : "plain launches:");
        for (int i = 0; i < n; i++) {
            const int blocks = seq[i].gx * seq[i].gy;
            int off = (h[i * 4096] - 0 + 8) % 8, ok = 1, cnt[8] = {0};
            for (int b = 0; b < blocks; b++) { cnt[h[i * 4096 + b] & 7]++; if (h[i * 4096 + b] != (b + off) % 8) ok = 0; }
            printf("  %-16s grid %4d x %2d: ", seq[i].name, seq[i].gx, seq[i].gy);
            if (ok) printf("xcc == (lin + %d) %% 8\n", off); else { printf("irregular; first 16:"); for (int b = 0; b < 16; b++) printf(" %d", h[i * 4096 + b]); printf("  per-XCD counts:"); for (int x = 0; x < 8; x++) printf(" %d", cnt[x]); printf("\n"); }
        }
    }
    return 0;
}
