// topk_device.h -- top-8 selection helpers shared by the EAGLE-2 tree-logic kernels (eagle_kernels.hip) and Token Recycle's table
// update (verify_kernels.hip): wave reductions on the DPP network and the 256-thread block top-8.  Order everywhere: value
// descending, index ascending among equal values (torch.topk's order on tie-free rows); index 0x7fffffff = "nothing".
#pragma once
#include <hip/hip_runtime.h>

#define E2_K 8
__device__ __forceinline__ bool e2_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// wave-wide max / min on the DPP network (row_shr 1, 2, 4, 8 inside each row of 16 lanes, row_bcast:15 / :31 across the rows; the
// result sits in lane 63): ~8 VALU instructions where a __shfl_xor butterfly is 6 dependent LDS-crossbar round trips.  All 64
// lanes must be active.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int e2_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ float e2_wave_max(float x) {
    const int ninf = __builtin_bit_cast(int, -INFINITY);
#define E2_STEP(CTRL, MASK) x = fmaxf(x, __builtin_bit_cast(float, e2_dpp<CTRL, MASK>(ninf, __builtin_bit_cast(int, x))))
    E2_STEP(0x111, 0xf); E2_STEP(0x112, 0xf); E2_STEP(0x114, 0xf); E2_STEP(0x118, 0xf); E2_STEP(0x142, 0xa); E2_STEP(0x143, 0xc);
#undef E2_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ int e2_wave_min(int x) {                 // non-negative values
#define E2_STEP(CTRL, MASK) x = min(x, e2_dpp<CTRL, MASK>(0x7fffffff, x))
    E2_STEP(0x111, 0xf); E2_STEP(0x112, 0xf); E2_STEP(0x114, 0xf); E2_STEP(0x118, 0xf); E2_STEP(0x142, 0xa); E2_STEP(0x143, 0xc);
#undef E2_STEP
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ float e2_wave_sum(float x) {
#define E2_STEP(CTRL, MASK) x += __builtin_bit_cast(float, e2_dpp<CTRL, MASK>(0, __builtin_bit_cast(int, x)))
    E2_STEP(0x111, 0xf); E2_STEP(0x112, 0xf); E2_STEP(0x114, 0xf); E2_STEP(0x118, 0xf); E2_STEP(0x142, 0xa); E2_STEP(0x143, 0xc);
#undef E2_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// the wave's best (value desc, index asc) of one (v, i) per lane; i = 0x7fffffff marks "nothing"
__device__ __forceinline__ void e2_wave_best(float v, int i, float &out_v, int &out_i) {
    out_v = e2_wave_max(v);
    out_i = e2_wave_min(v == out_v ? i : 0x7fffffff);
}

#define E2_SEG 4096                     // elements per stage-1 workgroup
#define E2_EPT 16                       // elements per thread (256 threads)
#define E2_MAXSPLIT 64

// The best 8 of a 256-thread workgroup's elements (N per thread, in registers), (value desc, index asc), left in res_v / res_i of
// wave 0.  Every wave first finds ITS best 8 without a barrier: 8 rounds of wave arg-max over the lanes' current best; a lane keeps
// its best and second best, so the winner usually just promotes its second (a rescan of its N elements, run by one lane while 63
// wait, only when it wins again: scripts/probes/top8_probe.hip -- the rescan was 60 % of a round, the block barrier + LDS exchange
// of a block-wide round another 25 %).  Then one barrier and wave 0 merges the 4 x 8 candidates, one per lane.
template <int N>
__device__ __forceinline__ void e2_block_top8(float (&v)[N], int (&id)[N], float (&res_v)[E2_K], int (&res_i)[E2_K], float *sv, int *si) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float b1v, b2v; int b1i, b2i;
    auto scan = [&]() {
        b1v = b2v = -INFINITY; b1i = b2i = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < N; q++) {
            if (e2_before(v[q], id[q], b1v, b1i)) { b2v = b1v; b2i = b1i; b1v = v[q]; b1i = id[q]; }
            else if (e2_before(v[q], id[q], b2v, b2i)) { b2v = v[q]; b2i = id[q]; }
        }
    };
    scan();
    bool have2 = true;
#pragma unroll 1
    for (int round = 0; round < E2_K; round++) {
        float wv; int wi;
        e2_wave_best(b1v, b1i, wv, wi);
        if (lane == 0) { sv[wave * E2_K + round] = wv; si[wave * E2_K + round] = wi; }
        if (b1i == wi && wi != 0x7fffffff) {               // indices are unique: this lane holds the winner
#pragma unroll
            for (int q = 0; q < N; q++) if (id[q] == wi) { v[q] = -INFINITY; id[q] = 0x7fffffff; }
            if (have2) { b1v = b2v; b1i = b2i; have2 = false; }
            else { scan(); have2 = true; }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float cv = lane < 4 * E2_K ? sv[lane] : -INFINITY; int ci = lane < 4 * E2_K ? si[lane] : 0x7fffffff;
        if (ci == 0x7fffffff) cv = -INFINITY;
#pragma unroll
        for (int round = 0; round < E2_K; round++) {
            float wv; int wi;
            e2_wave_best(cv, ci, wv, wi);
            res_v[round] = wv; res_i[round] = wi;
            if (ci == wi) { cv = -INFINITY; ci = 0x7fffffff; }
        }
    }
}


__device__ __forceinline__ float e2_f32(_Float16 x) { return (float)x; }
__device__ __forceinline__ float e2_f32(__bf16 x) { return (float)x; }
__device__ __forceinline__ float e2_f32(float x) { return x; }

// the 16 elements of a 256-thread workgroup's thread out of its 4096-element segment [seg0, seg0 + 4096) of row x: 16-byte vector
// loads when the row is 16-byte aligned, coalesced across the workgroup; elements past `vocab` read as (-inf, nothing)
template <typename T>
__device__ __forceinline__ void e2_load_segment(const T *__restrict__ x, long long vocab, long long seg0, float (&v)[E2_EPT], int (&id)[E2_EPT]) {
    constexpr int VEC = 16 / sizeof(T), NV = E2_EPT / VEC;
    const int tid = threadIdx.x;
    const bool vec_ok = ((((size_t)x) & 15) == 0);
    if (vec_ok) {
        uint4 raw[NV];
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const long long e0 = seg0 + ((long long)u * 256 + tid) * VEC;
            if (e0 + VEC <= vocab) raw[u] = *reinterpret_cast<const uint4 *>(x + e0);
        }
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const long long e0 = seg0 + ((long long)u * 256 + tid) * VEC;
            const T *e = reinterpret_cast<const T *>(&raw[u]);
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                const long long g = e0 + q;
                const bool in = g < vocab;
                v[u * VEC + q] = !in ? -INFINITY : (e0 + VEC <= vocab ? e2_f32(e[q]) : e2_f32(x[g]));
                id[u * VEC + q] = in ? (int)g : 0x7fffffff;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < E2_EPT; q++) {
            const long long g = seg0 + (long long)q * 256 + tid;
            const bool in = g < vocab;
            v[q] = in ? e2_f32(x[g]) : -INFINITY; id[q] = in ? (int)g : 0x7fffffff;
        }
    }
}
