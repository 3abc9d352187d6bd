// ingest_probe.hip -- what does a CU's vector-memory path ingest when a weight stream from HBM and an L2-resident activation tile share it?
// ("show whether the ~37 GB/s per CU whatever the source figure is a hardware limit".)  One workgroup of 512 threads per
// CU, 256 workgroups, no MFMA, no LDS reads -- only the loads of the 64-row weight-streaming GEMM (csrc/gemm_kernels.hip):
//   per chunk a workgroup reads 64 KiB of its PRIVATE weight slice (nt, 1 KiB per wave-instruction, HBM) and `a_kb` KiB of a SHARED
//   activation tile that every workgroup reads (L2-resident after the first touch; 32 KiB per chunk = the 64-row tile, 8 = the 16-row tile).
//   mode 0: weights only            mode 1: + A through registers, every wave loads its share (the shipped kernel's issue pattern)
//   mode 2: + A by LDS-DMA (global_load_lds, what the kernel does)      mode 3: + A by TWO dedicated waves, the other six stream the weights
//   mode 4: A only (what the L2 -> CU path does alone)                  mode 5: like 1 with the A loads issued BEFORE the chunk's weight loads
// usage: ingest_probe <mode> <a_kb per chunk> <chunks per workgroup> <depth> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4 *p) {           // the nt policy bit of the GEMM's weight loads
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 1) void k_ingest(const uint4 *__restrict__ W, const uint4 *__restrict__ A, int a_units /* 16-B units of A per chunk */, int a_total_units,
                                                   int chunks, unsigned *sink) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const uint4 *wp = W + (size_t)blockIdx.x * chunks * 4096;          // 64 KiB = 4096 units per chunk
    unsigned acc = 0;
    uint4 wr[DEPTH][8];
    uint4 ar[DEPTH][4];
    const bool w_wave = MODE != 4 && (MODE != 3 || w < 6);
    const bool a_wave = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || (MODE == 3 && w >= 6);
    auto issue_w = [&](int d, int c) {
        if (!w_wave) return;
        if (MODE == 3) {                                               // six waves share the chunk: 4096 units / 384 lanes -> 11 loads each (the last partly idle)
#pragma unroll
            for (int j = 0; j < 8; j++) { const int u = (w * 64 + l) + 384 * j; wr[d][j] = nt_load(wp + (size_t)c * 4096 + (u < 4096 ? u : 0)); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) wr[d][j] = nt_load(wp + (size_t)c * 4096 + 512 * j + tid);
        }
    };
    auto issue_a = [&](int d, int c) {
        if (!a_wave || a_units == 0) return;
        const int per_wave_inst = 64;                                  // units per wave-instruction
        const int a0 = (int)(((long long)c * a_units) % a_total_units);
        if (MODE == 3) {                                               // two waves load the whole A chunk
            const int lanes = 128, mine = (w - 6) * 64 + l;
#pragma unroll
            for (int j = 0; j < 4; j++) { const int u = mine + lanes * j; if (u < a_units) ar[d][j] = A[(a0 + u) % a_total_units]; }
            // (a_units = 2048 for 32 KiB: 16 loads per lane would be needed; 4 cover 512 units -> loop)
            for (int u = mine + lanes * 4; u < a_units; u += lanes) { const uint4 v = A[(a0 + u) % a_total_units]; acc ^= v.x; }
        } else if (MODE == 2) {
            for (int u0 = w * per_wave_inst; u0 < a_units; u0 += 512) {
                const uint4 *src = A + (a0 + u0 + l) % a_total_units;
                uint4 *dst = lds + (size_t)d * 2048 + u0;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { const int u = tid + 512 * j; if (u < a_units) ar[d][j] = A[(a0 + u) % a_total_units]; }
        }
    };
    auto consume = [&](int d) {
        if (w_wave) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= wr[d][j].x ^ wr[d][j].w;
        }
        if (a_wave && MODE != 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc ^= ar[d][j].y;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
        for (int j = 0; j < 4; j++) ar[d][j] = make_uint4(0, 0, 0, 0);
        if (d < chunks) { if (MODE == 5) { issue_a(d, d); issue_w(d, d); } else { issue_w(d, d); issue_a(d, d); } }
    }
    for (int c = 0; c < chunks; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (c + d >= chunks) break;
            consume(d);                                                // (the compiler waits for exactly this chunk's registers)
            if (MODE == 2) __builtin_amdgcn_s_waitcnt(0);              // LDS-DMA has no register to wait on: drain (pessimistic for mode 2)
            if (c + d + DEPTH < chunks) { if (MODE == 5) { issue_a(d, c + d + DEPTH); issue_w(d, c + d + DEPTH); } else { issue_w(d, c + d + DEPTH); issue_a(d, c + d + DEPTH); } }
        }
    }
    if (MODE == 2) acc ^= lds[tid].x;
    if (acc == 0x9E3779B9u) *sink = acc;
}

template <int MODE>
static int run(int a_kb, int chunks, int depth, int iters) {
    const int G = 256;
    const size_t w_bytes = (size_t)G * chunks * 65536;
    const int a_units = a_kb * 64, a_total_units = 512 * 64;           // the shared tile: 512 KiB (64 rows x 4096 k x 2 B)
    uint4 *W[3], *A; unsigned *sink;
    for (int i = 0; i < 3; i++) { CHK(hipMalloc(&W[i], w_bytes)); CHK(hipMemset(W[i], i + 1, w_bytes)); }
    CHK(hipMalloc(&A, (size_t)a_total_units * 16)); CHK(hipMemset(A, 7, (size_t)a_total_units * 16));
    CHK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int it) {
        const size_t lds = MODE == 2 ? (size_t)depth * 32768 : 16;
        if (depth == 2) hipLaunchKernelGGL((k_ingest<MODE, 2>), dim3(G), dim3(512), lds, 0, W[it % 3], A, a_units, a_total_units, chunks, sink);
        else hipLaunchKernelGGL((k_ingest<MODE, 3>), dim3(G), dim3(512), lds, 0, W[it % 3], A, a_units, a_total_units, chunks, sink);
    };
    if (MODE == 2) {
        CHK(hipFuncSetAttribute((const void *)k_ingest<MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
        CHK(hipFuncSetAttribute((const void *)k_ingest<MODE, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
    }
    launch(0); CHK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int it = 0; it < iters; it++) launch(it + 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double wb = MODE == 4 ? 0.0 : (double)w_bytes, ab = (double)G * chunks * a_units * 16.0;
    printf("mode %d a_kb %d chunks %d depth %d: %.2f us/launch; weights %.1f MB at %.2f TB/s; A %.1f MB (L2) ; per CU ingest %.1f GB/s (W %.1f + A %.1f)\n", MODE, a_kb, chunks, depth,
           ms * 1e3, wb / 1e6, wb / ms / 1e9, ab / 1e6, (wb + ab) / G / ms / 1e6, wb / G / ms / 1e6, ab / G / ms / 1e6);
    return 0;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, a_kb = argc > 2 ? atoi(argv[2]) : 32, chunks = argc > 3 ? atoi(argv[3]) : 11, depth = argc > 4 ? atoi(argv[4]) : 2;
    const int iters = argc > 5 ? atoi(argv[5]) : 20;
    switch (mode) {
    case 0: return run<0>(a_kb, chunks, depth, iters);
    case 1: return run<1>(a_kb, chunks, depth, iters);
    case 2: return run<2>(a_kb, chunks, depth, iters);
    case 3: return run<3>(a_kb, chunks, depth, iters);
    case 4: return run<4>(a_kb, chunks, depth, iters);
    default: return run<5>(a_kb, chunks, depth, iters);
    }
}
