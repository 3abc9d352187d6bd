// gemm_kernels.hip -- weight-streaming skinny GEMM of the verify forward for gfx950:  out[m][n] = sum_k A[m][k] * W[n][k]
// with m <= 64 draft rows, W = an HF nn.Linear weight [N, K], re-tiled once at load time (samd_gemm_pack_weights) so that
// every workgroup reads its share as one linear stream of 64 KiB blocks.
//
// At n <= 64 rows the verify forward is bound by reading the weights once (13.5 GB for Vicuna-7B): the kernel is a
// weight STREAM with a little MFMA attached.  What shapes it (scripts/hbm_probe.hip, profiles/r01_hbm_probe.md): the
// memory system retires ~48 G requests/s whatever their size, so every request must carry a full 128-byte line --
// each lane therefore owns 32 contiguous bytes of a weight row per 64-wide k block (4 lanes = 128 B of one row), and
// the MFMA k index is permuted accordingly (the A operand is read from LDS with the same permutation, so the product
// is unchanged).  PACKED LAYOUT (scripts/stream_probe.hip, profiles/r01_stream_sweep.log: reading the same bytes with this
// kernel's grid but the row-major matrix is 8-9 % slower, 79.6 vs 72.8 us for the four projections of a layer): block
// (tile t = 128 output columns, chunk c = 256 k) is 64 KiB contiguous at ((t * K/256 + c) * 4096) uint4 units; inside it
// unit (2b + j) * 512 + tid holds W[128 t + 16 w + n][256 c + 64 b + 16 g + 8 j .. +7] for tid = 64 w + 16 g + n -- exactly
// the order the lanes consume, so one wave-instruction is 1 KiB contiguous.  A (the activations, <= 512 KB, L2 resident) is staged per workgroup through LDS in 256-wide k chunks,
// double buffered; the weights go HBM -> VGPR -> MFMA one chunk ahead.  (Measured and dropped: a 16-row variant that stages the
// whole A slice once and streams without per-chunk barriers -- 159 vs 141 us per layer-set, the serial staging costs more.
// Ablation at 16 rows: with the MFMAs and the LDS traffic compiled out (weight stream only) the five projections of a layer
// take 137 us vs 141 us for the real kernel, and dropping the barriers changes nothing: the kernel runs at what separate
// launches of 33-180 MB can stream (4.0-5.2 TB/s incl. ramp-up and tail); the rest is launch granularity, not the kernel body.)  Split-K partial sums are written as fp32 and
// summed by the consuming kernel (rmsnorm+residual, rope, silu*up), so the split costs no extra launch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "samd_common.h"

#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct GF16 { typedef _Float16 elem; typedef half8 vec8;
    static __device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
struct GBF16 { typedef __bf16 elem; typedef bf16x8 vec8;
    static __device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };

#define GEMM_KC 256                    // k elements per chunk
#define GEMM_WAVES 8                   // waves per workgroup; each wave owns 16 output columns
#define GEMM_COLS (16 * GEMM_WAVES)    // output columns per workgroup

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <typename TT, int RT>
__global__ __launch_bounds__(64 * GEMM_WAVES, GEMM_WAVES / 2) void k_gemm_skinny(const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ W,
                                                        float *__restrict__ partial, typename TT::elem *__restrict__ out,
                                                        int K, int N, int n_chunks, int n_splits) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    constexpr int R = 16 * RT;
    constexpr int NT = 64 * GEMM_WAVES;
    constexpr int XV = (R * 32) / NT;              // 16-byte units per thread to stage one A chunk (R rows x 32 units)
    // A tile: [R rows][32 units of 16 B], unit u of row r stored at position u ^ (r & 15): the 16 rows that one
    // ds_read_b128 wave-instruction touches (same unit, rows m..m+15) land in 16 different 16-byte slots -> no bank
    // conflict, and rows stay contiguous so the tile can be filled by LDS-DMA (global_load_lds, 1 KiB per wave-instruction,
    // no VGPR round trip; the swizzle goes on the SOURCE address).
    __shared__ __attribute__((aligned(1024))) E xs[2][R][GEMM_KC];

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int n0 = blockIdx.x * GEMM_COLS + 16 * w;
    const int split = blockIdx.y;
    const int c0 = (int)((long long)split * n_chunks / n_splits), c1 = (int)((long long)(split + 1) * n_chunks / n_splits);
    const uint4 *wtile = reinterpret_cast<const uint4 *>(W) + (size_t)blockIdx.x * n_chunks * 4096 + tid;   // packed: see header

    floatx4 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; mt++) acc[mt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    uint4 wa[4][2], wb[4][2];
    auto load_w = [&](uint4 (&dst)[4][2], int c) {
        const uint4 *p = wtile + (size_t)c * 4096;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            dst[b][0] = p[512 * (2 * b)];
            dst[b][1] = p[512 * (2 * b + 1)];
        }
    };
    auto stage_x = [&](int c, int buf) {          // asynchronous: lands in LDS, counted by vmcnt
#pragma unroll
        for (int i = 0; i < XV; i++) {
            const int slot = tid + NT * i, row = slot >> 5, pos = slot & 31, unit = pos ^ (row & 15);
            const E *src = A + (size_t)row * K + (size_t)c * GEMM_KC + 8 * unit;
            E *dst = &xs[buf][0][0] + (size_t)(NT * i + 64 * w) * 8;       // wave-uniform base; the hardware adds lane * 16 B
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
    };
    // one phase = the MFMAs of chunk c on `cur` (weights loaded one phase earlier) and LDS buffer `buf`, while the
    // weights and the A tile of chunk c+1 are in flight; two alternating phases so the weight registers are never copied
    auto phase = [&](uint4 (&cur)[4][2], uint4 (&nxt)[4][2], int c, int buf) {
        const bool more = c + 1 < c1;
        if (more) { load_w(nxt, c + 1); stage_x(c + 1, buf ^ 1); }
#pragma unroll
        for (int b = 0; b < 4; b++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const V8 bf = __builtin_bit_cast(V8, cur[b][j]);
                const int unit = (8 * b + 2 * g + j) ^ n;                    // rows 16 mt + n: (row & 15) == n
#pragma unroll
                for (int mt = 0; mt < RT; mt++) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(&xs[buf][16 * mt + n][8 * unit]);
                    acc[mt] = TT::mfma(__builtin_bit_cast(V8, raw), bf, acc[mt]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // next chunk's weights + A tile have landed
        __syncthreads();
    };
    if (c0 < c1) {
        load_w(wa, c0);
        stage_x(c0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = c0; c < c1; c += 2) {
            phase(wa, wb, c, 0);
            if (c + 1 < c1) phase(wb, wa, c + 1, 1);
        }
    }
    // C layout of mfma_16x16: lane holds rows 4g + r of column n
#pragma unroll
    for (int mt = 0; mt < RT; mt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = 16 * mt + 4 * g + r;
            if (out) out[(size_t)m * N + n0 + n] = (E)acc[mt][r];
            else partial[((size_t)split * R + m) * N + n0 + n] = acc[mt][r];
        }
    }
}

// row-major [N][K] (2-byte elements) -> packed blocks; one thread moves one 16-byte unit
__global__ __launch_bounds__(256) void k_gemm_pack(const uint4 *__restrict__ W, uint4 *__restrict__ out, int N, int K) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;          // destination unit
    const long long total = (long long)N * K / 8;
    if (u >= total) return;
    const int n_chunks = K / GEMM_KC;
    const long long blk = u >> 12;
    const int in = (int)(u & 4095), jj = in >> 9, tid = in & 511, w = tid >> 6, g = (tid >> 4) & 3, n = tid & 15;
    const int t = (int)(blk / n_chunks), c = (int)(blk % n_chunks), b = jj >> 1, j = jj & 1;
    const long long row = 128LL * t + 16 * w + n, col = 256LL * c + 64 * b + 16 * g + 8 * j;
    out[u] = W[(row * K + col) / 8];
}

extern "C" {

// choose the split-K factor: enough workgroups to cover the chip twice, never more splits than chunks, and few enough
// that the fp32 partials (written here, read by the consumer) stay a small fraction of the weight bytes
int samd_gemm_splits(int32_t N, int32_t K, int32_t rows_pad) {
    const int cols = N / GEMM_COLS, chunks = K / GEMM_KC;
    int cap = rows_pad <= 32 ? 8 : 4;
    if (const char *e = getenv("SAMD_GEMM_SPLIT_CAP")) cap = atoi(e);
    if (const char *e = getenv("SAMD_GEMM_SPLITS")) { int v = atoi(e); return v < 1 ? 1 : (v > chunks ? chunks : v); }
    int s = 1;
    while (cols * s < 512 && s * 2 <= chunks && s * 2 <= cap) s *= 2;
    return s;
}

int64_t samd_gemm_workspace(int32_t rows_pad, int32_t N, int32_t splits) { return (int64_t)splits * rows_pad * N * 4; }

// out (dtype, [rows_pad][N]) when splits == 1, else fp32 partials [splits][rows_pad][N] in d_partial.
// rows_pad in {16, 32, 64}; A must hold rows_pad rows (pad rows are read, their products land in pad rows).
int samd_gemm_pack_weights(const void *d_W, void *d_packed, int32_t N, int32_t K, void *stream) {
    if (!d_W || !d_packed || d_W == d_packed || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC || K % GEMM_KC != 0) {
        samd_set_error("samd_gemm_pack_weights: needs N %% 128 == 0, K %% 256 == 0 and distinct buffers"); return SAMD_E_INVALID;
    }
    const long long units = (long long)N * K / 8;
    hipLaunchKernelGGL(k_gemm_pack, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_W, (uint4 *)d_packed, N, K);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_skinny(const void *d_A, const void *d_W, int32_t rows_pad, int32_t N, int32_t K, int32_t splits, float *d_partial,
                     void *d_out, int32_t dtype, void *stream) {
    if (!d_A || !d_W || (rows_pad != 16 && rows_pad != 32 && rows_pad != 64) || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC ||
        K % GEMM_KC != 0 || splits < 1 || splits > K / GEMM_KC || (splits == 1 ? !d_out : !d_partial) || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_skinny: unsupported shape (rows 16/32/64, N %% 128 == 0, K %% 256 == 0) or null pointer"); return SAMD_E_INVALID;
    }
    const dim3 grid(N / GEMM_COLS, splits), block(64 * GEMM_WAVES);
    hipStream_t st = (hipStream_t)stream;
    const int chunks = K / GEMM_KC;
    void *out = splits == 1 ? d_out : nullptr;
#define GO(TT, RT) hipLaunchKernelGGL((k_gemm_skinny<TT, RT>), grid, block, 0, st, (const TT::elem *)d_A, (const TT::elem *)d_W, d_partial, (TT::elem *)out, K, N, chunks, splits)
    if (dtype == SAMD_F16) { if (rows_pad == 16) GO(GF16, 1); else if (rows_pad == 32) GO(GF16, 2); else GO(GF16, 4); }
    else { if (rows_pad == 16) GO(GBF16, 1); else if (rows_pad == 32) GO(GBF16, 2); else GO(GBF16, 4); }
#undef GO
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
