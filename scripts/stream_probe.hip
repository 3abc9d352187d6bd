// stream_probe.hip -- why does the skinny GEMM stream weights slower than a linear sweep?  Reads an [N][K] fp16 matrix
// (no math, no LDS) with different address patterns and grid shapes; cold = 6 rotating buffers.
//   mode 0: linear sweep, G workgroups x 256 threads, 8 x 16 B in flight per lane (the Infinity-Cache warmer)
//   mode 1: the GEMM's pattern on the row-major matrix: grid (N/128, S) x 512 threads, lane = 32 B of a row per 64-k block,
//           chunk = 256 k, DEPTH chunks in flight
//   mode 2: the GEMM's grid on a TILED matrix: workgroup (tile, split) reads its chunks as contiguous 64 KiB blocks
//           (wave-instruction = 1 KiB contiguous)
//   mode 3: tiled + persistent: G workgroups x 512 threads walk the (tile, chunk) blocks round-robin
// usage: stream_probe <mode> <N> <K> <S or G> <depth> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_linear(const uint4 *__restrict__ p, long long n16, unsigned *sink) {
    const long long stride = (long long)gridDim.x * 256;
    unsigned acc = 0;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].w;
    }
    for (; i < n16; i += stride) acc ^= p[i].x;
    if (acc == 0x9E3779B9u) *sink = acc;
}

template <int DEPTH, bool TILED>
__global__ __launch_bounds__(512, 4) void k_gemmlike(const _Float16 *__restrict__ W, int K, int N, int n_chunks, int n_splits, unsigned *sink) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int split = blockIdx.y;
    const int c0 = (int)((long long)split * n_chunks / n_splits), c1 = (int)((long long)(split + 1) * n_chunks / n_splits);
    unsigned acc = 0;
    uint4 buf[DEPTH][8];
    auto load = [&](uint4 (&d)[8], int c) {
        if (TILED) {
            const uint4 *p = reinterpret_cast<const uint4 *>(W) + ((size_t)blockIdx.x * n_chunks + c) * 4096 + tid;     // 64 KiB block
#pragma unroll
            for (int j = 0; j < 8; j++) d[j] = p[512 * j];
        } else {
            const _Float16 *p = W + (size_t)(blockIdx.x * 128 + 16 * w + n) * K + 16 * g + (size_t)c * 256;
#pragma unroll
            for (int b = 0; b < 4; b++) { d[2 * b] = *reinterpret_cast<const uint4 *>(p + 64 * b); d[2 * b + 1] = *reinterpret_cast<const uint4 *>(p + 64 * b + 8); }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) if (c0 + d < c1) load(buf[d], c0 + d);
    for (int c = c0; c < c1; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (c + d >= c1) break;
            if (c + d + DEPTH - 1 < c1) load(buf[(d + DEPTH - 1) % DEPTH], c + d + DEPTH - 1);
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= buf[d][j].x ^ buf[d][j].w;
        }
    }
    if (acc == 0x9E3779B9u) *sink = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(512, 4) void k_tiled_persistent(const uint4 *__restrict__ W, long long n_blocks, unsigned *sink) {
    const int tid = threadIdx.x;
    unsigned acc = 0;
    uint4 buf[DEPTH][8];
    auto load = [&](uint4 (&d)[8], long long b) {
        const uint4 *p = W + b * 4096 + tid;
#pragma unroll
        for (int j = 0; j < 8; j++) d[j] = p[512 * j];
    };
    const long long G = gridDim.x;
    long long b = blockIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) if (b + d * G < n_blocks) load(buf[d], b + d * G);
    for (; b < n_blocks; b += DEPTH * G) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (b + d * G >= n_blocks) break;
            if (b + (d + DEPTH - 1) * G < n_blocks) load(buf[(d + DEPTH - 1) % DEPTH], b + (d + DEPTH - 1) * G);
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= buf[d][j].x ^ buf[d][j].w;
        }
    }
    if (acc == 0x9E3779B9u) *sink = acc;
}

int main(int argc, char **argv) {
    if (argc < 6) { printf("usage: stream_probe <mode> <N> <K> <S|G> <depth> [iters]\n"); return 2; }
    const int mode = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), SG = atoi(argv[4]), depth = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 30;
    const size_t bytes = (size_t)N * K * 2;
    const int NB = 6;
    char *base; unsigned *sink;
    CHK(hipMalloc((void **)&base, bytes * NB)); CHK(hipMemset(base, 1, bytes * NB)); CHK(hipMalloc((void **)&sink, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto launch = [&](int it) {
        const void *p = base + bytes * (it % NB);
        const int chunks = K / 256;
        if (mode == 0) hipLaunchKernelGGL(k_linear, dim3(SG), dim3(256), 0, 0, (const uint4 *)p, (long long)(bytes / 16), sink);
        else if (mode == 1 || mode == 2) {
            dim3 grid(N / 128, SG);
#define GO(D, T) hipLaunchKernelGGL((k_gemmlike<D, T>), grid, dim3(512), 0, 0, (const _Float16 *)p, K, N, chunks, SG, sink)
            if (mode == 1) { if (depth == 1) GO(1, false); else if (depth == 2) GO(2, false); else GO(3, false); }
            else { if (depth == 1) GO(1, true); else if (depth == 2) GO(2, true); else GO(3, true); }
#undef GO
        } else {
            const long long nb = (long long)(bytes / 65536);
            if (depth == 1) hipLaunchKernelGGL(k_tiled_persistent<1>, dim3(SG), dim3(512), 0, 0, (const uint4 *)p, nb, sink);
            else if (depth == 2) hipLaunchKernelGGL(k_tiled_persistent<2>, dim3(SG), dim3(512), 0, 0, (const uint4 *)p, nb, sink);
            else hipLaunchKernelGGL(k_tiled_persistent<3>, dim3(SG), dim3(512), 0, 0, (const uint4 *)p, nb, sink);
        }
    };
    for (int it = 0; it < 6; it++) launch(it);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) launch(it);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("mode %d N=%5d K=%5d S/G=%4d depth=%d: %7.1f us  %6.2f TB/s\n", mode, N, K, SG, depth, ms * 1e3, bytes / ms / 1e9);
    return 0;
}
