// top8_probe.hip -- where does a round of the block top-8 (csrc/eagle_kernels.hip e2_block_top8) spend its time?  Variants of
// one 256-thread workgroup over 16 register elements per thread, timed with s_memrealtime (100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/top8_probe scripts/probes/top8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ bool before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ float wave_max(float x) {
    const int ninf = __builtin_bit_cast(int, -INFINITY);
#define STEP(CTRL, MASK) x = fmaxf(x, __builtin_bit_cast(float, dpp<CTRL, MASK>(ninf, __builtin_bit_cast(int, x))))
    STEP(0x111, 0xf); STEP(0x112, 0xf); STEP(0x114, 0xf); STEP(0x118, 0xf); STEP(0x142, 0xa); STEP(0x143, 0xc);
#undef STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ int wave_min(int x) {
#define STEP(CTRL, MASK) x = min(x, dpp<CTRL, MASK>(0x7fffffff, x))
    STEP(0x111, 0xf); STEP(0x112, 0xf); STEP(0x114, 0xf); STEP(0x118, 0xf); STEP(0x142, 0xa); STEP(0x143, 0xc);
#undef STEP
    return __builtin_amdgcn_readlane(x, 63);
}
// MODE bit 0: skip the winner's rescan; bit 1: skip barrier + LDS exchange (wave-local result); bit 2: butterfly shuffles instead of DPP
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, long long *ticks) {
    __shared__ float sv[8]; __shared__ int si[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[16]; int id[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { v[q] = in[q * 256 + tid]; id[q] = q * 256 + tid; }
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 16; q++) if (before(v[q], id[q], bv, bi)) { bv = v[q]; bi = id[q]; }
    __syncthreads();
    const long long t0 = wall_clock64();
    float res[8];
    if (MODE & 16) {                                       // odd-even transposition sort of the thread's 16 (value desc, index asc), all threads in parallel
#pragma unroll
        for (int pass = 0; pass < 16; pass++)
#pragma unroll
            for (int q = pass & 1; q + 1 < 16; q += 2)
                if (before(v[q + 1], id[q + 1], v[q], id[q])) { const float tv = v[q]; const int ti = id[q]; v[q] = v[q + 1]; id[q] = id[q + 1]; v[q + 1] = tv; id[q + 1] = ti; }
        bv = v[0]; bi = id[0];
    }
#pragma unroll 1
    for (int rr = 0; rr < ((MODE & 8) ? 8 : 1); rr++)
#pragma unroll
    for (int r2 = 0; r2 < ((MODE & 8) ? 1 : 8); r2++) {
        const int round = (MODE & 8) ? rr : r2;
        float wv; int wi;
        if (MODE & 4) {
            wv = bv; wi = bi;
            for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(wv, o); const int oi = __shfl_xor(wi, o); if (before(ov, oi, wv, wi)) { wv = ov; wi = oi; } }
        } else { wv = wave_max(bv); wi = wave_min(bv == wv ? bi : 0x7fffffff); }
        float gv = wv; int gi = wi;
        if (!(MODE & 2)) {
            const int slot = (round & 1) * 4;
            if (lane == 0) { sv[slot + wave] = wv; si[slot + wave] = wi; }
            __syncthreads();
            gv = sv[slot]; gi = si[slot];
#pragma unroll
            for (int k2 = 1; k2 < 4; k2++) { const float ov = sv[slot + k2]; const int oi = si[slot + k2]; if (before(ov, oi, gv, gi)) { gv = ov; gi = oi; } }
        }
        if (MODE & 8) { if (tid == 0) out[8 + round] = gv; } else res[round] = gv;
        if (bi == gi && gi != 0x7fffffff) {
            if (MODE & 1) { bv = -INFINITY; bi = 0x7fffffff; }
            else if (MODE & 16) {
#pragma unroll
                for (int q = 0; q < 15; q++) { v[q] = v[q + 1]; id[q] = id[q + 1]; }
                v[15] = -INFINITY; id[15] = 0x7fffffff; bv = v[0]; bi = id[0];
            }
            else {
                bv = -INFINITY; bi = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < 16; q++) { if (id[q] == gi) { v[q] = -INFINITY; id[q] = 0x7fffffff; } if (before(v[q], id[q], bv, bi)) { bv = v[q]; bi = id[q]; } }
            }
        }
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { ticks[0] = t1 - t0; if (!(MODE & 8)) for (int r = 0; r < 8; r++) out[r] = res[r]; else for (int r = 0; r < 8; r++) out[r] = out[8 + r]; }
}
int main() {
    float *in, *out; long long *ticks;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 128); hipMalloc(&ticks, 8);
    float h[4096]; for (int i = 0; i < 4096; i++) h[i] = sinf(i * 12.9898f) * 43758.5453f - floorf(sinf(i * 12.9898f) * 43758.5453f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
#define RUN(M) { long long t[3]; float o[8]; for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, in, out, ticks); hipDeviceSynchronize(); \
    hipMemcpy(&t[rep], ticks, 8, hipMemcpyDeviceToHost); } hipMemcpy(o, out, 32, hipMemcpyDeviceToHost); \
    printf("mode %2d: first launch %5lld, then %5lld %5lld ticks (x10 ns) for 8 rounds; top = %.5f %.5f ... %.5f\n", M, t[0], t[1], t[2], o[0], o[1], o[7]); }
    RUN(0) RUN(8) RUN(16) RUN(24) RUN(1) RUN(3) RUN(18) RUN(26)
    return 0;
}
