// norm_probe.hip -- what would spreading the residual-add + RMSNorm kernel over more workgroups buy?  (a) today's shape: one
// 512-thread workgroup per row (16 rows) reads x, the weight and 8 fp32 split-K partials, block-reduces, writes x and h;
// (b) the same bytes with 4 / 8 workgroups per row and only a per-chunk sum of squares written (what a "deferred 1/rms" scheme
// would launch instead).  200 back-to-back launches captured in one hipGraph, HIP events around the replay.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/norm_probe scripts/probes/norm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define H 4096
#define NP 8
__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}
// chunks = workgroups per row; FULL = normalise inside the kernel (needs chunks == 1)
template <bool FULL>
__global__ void k_norm(_Float16 *x, const float *part, const _Float16 *w, _Float16 *out, float *ss, int chunks, int rows_pad) {
    __shared__ float red[16];
    const int row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int c = (chunk * blockDim.x + threadIdx.x) * 8;
    const size_t off = (size_t)row * H + c;
    h8 a = *reinterpret_cast<const h8 *>(x + off);
    const h8 wv = *reinterpret_cast<const h8 *>(w + c);
    f4 p[NP][2];
#pragma unroll
    for (int s = 0; s < NP; s++) { p[s][0] = *reinterpret_cast<const f4 *>(part + (size_t)s * rows_pad * H + off); p[s][1] = *reinterpret_cast<const f4 *>(part + (size_t)s * rows_pad * H + off + 4); }
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < NP; s++) d += p[s][j >> 2][j & 3];
        a[j] = (_Float16)((float)a[j] + (float)(_Float16)d);
        sq += (float)a[j] * (float)a[j];
    }
    *reinterpret_cast<h8 *>(x + off) = a;
    const float tot = block_sum(sq, red);
    h8 o;
    if (FULL) {
        const float rs = rsqrtf(tot / H + 1e-5f);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)wv[j] * (float)(_Float16)((float)a[j] * rs));
    } else {
        if (threadIdx.x == 0) ss[row * chunks + chunk] = tot;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)wv[j] * (float)a[j]);
    }
    *reinterpret_cast<h8 *>(out + off) = o;
}
int main() {
    const int rows = 16, rows_pad = 16;
    _Float16 *x, *w, *out; float *part, *ss;
    hipMalloc(&x, rows * H * 2); hipMalloc(&w, H * 2); hipMalloc(&out, rows * H * 2); hipMalloc(&part, (size_t)NP * rows_pad * H * 4); hipMalloc(&ss, 4096);
    hipMemset(x, 0, rows * H * 2); hipMemset(w, 0, H * 2); hipMemset(part, 0, (size_t)NP * rows_pad * H * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 4; variant++) {
        const int chunks = variant == 0 ? 1 : (variant == 1 ? 2 : (variant == 2 ? 4 : 8));
        const int threads = H / 8 / chunks;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; i++) {
            if (chunks == 1) hipLaunchKernelGGL(k_norm<true>, dim3(rows), dim3(threads), 0, st, x, part, w, out, ss, 1, rows_pad);
            else hipLaunchKernelGGL(k_norm<false>, dim3(rows * chunks), dim3(threads), 0, st, x, part, w, out, ss, chunks, rows_pad);
        }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%d workgroup(s) per row x %4d threads (%s): %.2f us per launch\n", chunks, threads, chunks == 1 ? "normalises in place" : "sum of squares per chunk", ms * 1000 / 1000);
    }
    return 0;
}
