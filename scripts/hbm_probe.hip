// hbm_probe.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of the SAM walk kernel
// (MI355X_MICROARCH.md, section HBM: "calibrate on a known byte count in your own access pattern").
//   mode 0: every lane reads one whole 64-byte line (4 x dwordx4) at a pseudo-random line index; every line of the
//           buffer is touched exactly once per launch  -> known bytes = lines x 64
//   mode 1: wide coalesced streaming read (16 B per lane, consecutive lanes consecutive addresses) of the same buffer
//   mode 2: like mode 0 but only the first 16 bytes of each line (one dwordx4)
//   mode 3: four consecutive lanes share one random line, 16 bytes each (one 64-byte request per lane quad)
//   mode 4: like mode 3 with eight lanes per random 128-byte block
//   mode 5: one lane reads 16 bytes at offset 0 AND 16 bytes at offset 64 of a random 128-byte block (are the two
//           halves of a 128-byte block one memory request or two?)
//   mode 6: 2^26 lanes each read 16 bytes of a pseudo-random 64-byte line inside a REGION of 2^<log2 lines> lines (working sets
//           that fit the 256 MiB Infinity Cache or the 32 MiB of L2: how fast are scattered reads that do not reach HBM?)
// usage: hbm_probe <mode> <log2 lines> <iters>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_lines(const int4 *__restrict__ base, unsigned long long n_lines, int mode, int *__restrict__ sink) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (mode == 6 ? (1ull << 26) : (mode == 4 ? n_lines / 2 : n_lines))) return;
    int acc = 0;
    if (mode == 3 || mode == 4) {
        const int per = mode == 3 ? 4 : 8;
        const unsigned long long n_blk = n_lines * 4 / per;
        for (int k = 0; k < per; k++) {               // `per` passes so that every lane quad/octet covers `per` blocks
            const unsigned long long g = (i / per) * per + k;
            if (g >= n_blk) break;
            const unsigned long long blk = (g * 0x9E3779B97F4A7C15ull) & (n_blk - 1);
            const int4 v = base[blk * per + (i % per)];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    } else if (mode == 6) {
        const unsigned long long line = ((i * 0x9E3779B97F4A7C15ull) >> 17) & (n_lines - 1);
        const int4 a = base[line * 4];
        acc = a.x ^ a.y ^ a.z ^ a.w;
    } else if (mode == 5) {
        const unsigned long long n_blk = n_lines / 2;
        if (i < n_blk) {
            const unsigned long long blk = (i * 0x9E3779B97F4A7C15ull) & (n_blk - 1);
            const int4 a = base[blk * 8], b = base[blk * 8 + 4];
            acc = a.x ^ a.w ^ b.y ^ b.z;
        }
    } else if (mode == 1) {
        for (int k = 0; k < 4; k++) { const int4 v = base[i + (unsigned long long)k * n_lines]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else {
        const unsigned long long line = (i * 0x9E3779B97F4A7C15ull) & (n_lines - 1);     // odd multiplier: a bijection
        const int4 *p = base + line * 4;
        const int4 a = p[0];
        acc = a.x ^ a.y ^ a.z ^ a.w;
        if (mode == 0) { const int4 b = p[1], c = p[2], d = p[3]; acc ^= b.x ^ b.y ^ c.z ^ c.w ^ d.x ^ d.w; }
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int lg = argc > 2 ? atoi(argv[2]) : 26;
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const unsigned long long n_lines = 1ull << lg;
    int4 *buf; int *sink;
    if (hipMalloc(&buf, n_lines * 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, n_lines * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned long long n_threads = (mode == 6) ? (1ull << 26) : ((mode == 3) ? n_lines : (mode == 4 ? n_lines / 2 : n_lines));
    const unsigned blocks = (unsigned)((n_threads + 255) / 256);
    hipLaunchKernelGGL(k_lines, dim3(blocks), dim3(256), 0, 0, buf, n_lines, mode, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(k_lines, dim3(blocks), dim3(256), 0, 0, buf, n_lines, mode, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    if (mode == 6) { printf("mode 6 region 2^%d lines (%.0f MiB): %.3f ms/launch, %.1f G reads/s\n", lg, (double)n_lines * 64 / 1048576.0, ms, (double)(1ull << 26) / ms / 1e6); return 0; }
    const double bytes = (double)n_lines * (mode == 2 ? 16 : 64);
    printf("mode %d lines 2^%d: %.3f ms/launch, requested bytes %.0f (%.1f GB/s); line bytes %.0f (%.1f GB/s)\n", mode, lg, ms, bytes,
           bytes / ms / 1e6, (double)n_lines * 64, (double)n_lines * 64 / ms / 1e6);
    return 0;
}
