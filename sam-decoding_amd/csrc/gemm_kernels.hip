// gemm_kernels.hip -- weight-streaming skinny GEMM of the verify forward for gfx950:  out[m][n] = sum_k A[m][k] * W[n][k]
// with m <= 64 draft rows, W = an HF nn.Linear weight [N, K], re-tiled once at load time (samd_gemm_pack_weights) so that
// every workgroup reads its share as one linear stream of 64 KiB blocks.
//
// At <= 64 rows the verify forward is bound by reading the weights once (13.5 GB for Vicuna-7B): the kernel is a weight
// STREAM with a little MFMA attached.  What shapes it (scripts/hbm_probe.hip, scripts/stream_probe.hip, profiles/):
//   * the memory system retires ~48 G requests/s whatever their size, so every request carries full lines: a lane loads
//     16 bytes, a wave-instruction 1 KiB contiguous;
//   * PACKED LAYOUT: block (tile t = 128 output columns, chunk c = 256 k) is 64 KiB contiguous at
//     ((t * K/256 + c) * 4096) uint4 units; inside it unit (2b + j) * 512 + tid holds
//     W[128 t + 16 w + n][256 c + 64 b + 16 g + 8 j .. +7] for tid = 64 w + 16 g + n -- exactly the order the lanes consume
//     (lane (g, n) of wave w feeds column 16 w + n; its 8-element vectors cover k = 64 b + 16 g + 8 j .. within each 64-wide
//     k block, and the A operand is read from LDS with the same k permutation, so the product is unchanged).  Reading the
//     same bytes with this grid from the row-major matrix is 8-9 % slower (79.6 vs 72.8 us for a layer's four projections);
//   * A (the activations, <= 1 MB, L2 resident) is staged per workgroup through LDS by LDS-DMA in 256-wide k chunks, three
//     buffers; the weights go HBM -> VGPR -> MFMA with TWO chunks in flight (see the pipeline comment in the kernel);
//   * one balanced wave of workgroups per launch (samd_gemm_splits); split-K partial sums are written as fp32 and added up
//     by the consuming kernel (rmsnorm + residual, rope), so a split costs no extra launch.
//   * the weight loads carry the nt (non-temporal) policy bit: -10 % on every projection launch (round 2; an earlier test through
//     __builtin_nontemporal_load on the pre-asm kernel had shown nothing);
// Measured and dropped: a 16-row variant that stages the whole A slice once (159 vs 141 us per layer),
// 16-wave workgroups, an intra-workgroup K split, an Infinity-Cache warmer on a side stream, producers of A inside the launch
// (DESIGN.md, K7).  Ablation: with the MFMAs and the LDS traffic compiled out the kernel is 3 % faster -- it runs at what
// separate launches of 33-180 MB can stream (4.1-5.9 TB/s incl. ramp-up and tail).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "samd_common.h"

#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

static int samd_cu_count() { return samd_device_cus(); }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GF16 { typedef _Float16 elem; typedef half8 vec8;
    static __device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
struct GBF16 { typedef __bf16 elem; typedef bf16x8 vec8;
    static __device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };

#define GEMM_KC 256                    // k elements per chunk
#define GEMM_WAVES 8                   // waves per workgroup; each wave owns 16 output columns
#define GEMM_COLS (16 * GEMM_WAVES)    // output columns per workgroup

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Diagnostic build only (-DSAMD_GEMM_ABLATE, scripts/r05_ablate.sh; never in the shipped library): k_gemm_pairs_silu with parts switched off at
// run time by SAMD_GEMM_ABL -- bit 0: no s_barrier per chunk (results wrong, timing only), bit 1: no LDS reads / MFMAs, bit 2: no A staging,
// bit 3: A fetched from rows 0..15 only (L1 hits instead of L2 traffic) -- to see where the 64-row tile loses its time (profiles/r05_wide_tile.md)
#ifdef SAMD_GEMM_ABLATE
__device__ int samd_abl_flag;
#endif

// EPI 0: out / fp32 partials as they are.  EPI 1 (splits == 1 only): the matrix is the MLP's gate|up pair with its rows
// interleaved in groups of 64 (tile t = gate columns 64t.. | up columns 64t..), and the epilogue writes
// silu(gate) * up [rows][N/2] -- LlamaMLP's activation without a launch, a 2N-wide intermediate or its re-read.
// DEPTH = weight chunks (+ their A tiles) in flight whenever a wave waits; DEPTH + 1 LDS buffers (dynamic LDS: 128 KiB at 64 rows).
// GM (round 6): W is GROUP-MAJOR (samd_gemm_pack_groups: column group gi = 16 columns, chunk c = 8 KiB contiguous at (gi * K/256 + c) * 512 units,
// unit (2b + j) * 64 + lane -- the layout k_gemm_cs_residual and k_gemm_pairs_silu stream) instead of 128-column tiles: wave w of tile t reads
// group 8 t + w as its own 1 KiB-per-instruction stream.  o_proj / down_proj then need ONE packed copy for every row bucket (the complete-sum
// kernels of <= 16 rows and this split-K kernel above them) instead of two: -4 GB of a 7B replica.
template <typename TT, int RT, int EPI, int DEPTH, bool GM = false>
__global__ __launch_bounds__(64 * GEMM_WAVES, RT >= 3 || DEPTH > 2 ? 2 : GEMM_WAVES / 2) void k_gemm_skinny(const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ W,
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
    constexpr int NB = DEPTH + 1;
    extern __shared__ __attribute__((aligned(1024))) char gemm_lds[];
    E (*xs)[R][GEMM_KC] = reinterpret_cast<E (*)[R][GEMM_KC]>(gemm_lds);

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int n0 = blockIdx.x * GEMM_COLS + 16 * w;
    const int split = blockIdx.y;
    const int c0 = (int)((long long)split * n_chunks / n_splits), c1 = (int)((long long)(split + 1) * n_chunks / n_splits);
    constexpr size_t WCH = GM ? 8192 : 65536, WU = GM ? 1024 : 8192;      // bytes of one (stream, chunk) block; of one (b, j) unit row inside it
    const int ws = GM ? __builtin_amdgcn_readfirstlane(GEMM_WAVES * (int)blockIdx.x + w) : (int)blockIdx.x;      // this wave's stream: its group / the tile
    const char *wtile = reinterpret_cast<const char *>(W) + (size_t)ws * n_chunks * WCH;           // packed: see header
    const uint32_t wlane = (uint32_t)(GM ? l : tid) * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)&xs[0][0][0];

    floatx4 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; mt++) acc[mt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // The weight loads, the LDS reads and the waits between them are hand-issued.  The compiler's wait-count pass is
    // path-insensitive and cannot tell which LDS buffer an LDS-DMA in flight writes: left to itself it drains every
    // outstanding load (vmcnt(0)) in front of each chunk's first LDS read and first MFMA, which leaves ONE chunk in flight.
    // Here TWO chunks (+ their A tiles) are in flight whenever a wave waits: memory ops retire in issue order, so "chunk c
    // has landed, chunk c+1 may still fly" is vmcnt(8 + XV).  scripts/stream_probe.hip: 18.3 vs 20.2 us for the QKV matrix.
    u32x4 wr[DEPTH][4][2];
    auto load_wb = [&](u32x4 (&dst)[4][2], int c, int b) {
        const char *p = wtile + (size_t)c * WCH;                     // wave-uniform -> SGPR base, one offset VGPR
#pragma unroll
        for (int j = 0; j < 2; j++)        // nt: every weight byte is read once, by one CU -- streamed past the caches (13.7 vs 15.1 us, 30.1 vs 33.3)
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst[b][j]) : "v"(wlane), "s"(p + WU * (2 * b + j)) : "memory");
    };
    auto stage_xi = [&](int c, int buf, int i) {   // asynchronous: lands in LDS, counted by vmcnt
        const int slot = tid + NT * i, row = slot >> 5, pos = slot & 31, unit = pos ^ (row & 15);
        const E *src = A + (size_t)row * K + (size_t)c * GEMM_KC + 8 * unit;
        E *dst = &xs[buf][0][0] + (size_t)(NT * i + 64 * w) * 8;       // wave-uniform base; the hardware adds lane * 16 B
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        asm volatile("" ::: "memory");
    };
    auto load_w = [&](u32x4 (&dst)[4][2], int c) {
#pragma unroll
        for (int b = 0; b < 4; b++) load_wb(dst, c, b);
    };
    auto stage_x = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < XV; i++) stage_xi(c, buf, i);
    };
    // wait for the oldest chunk in flight, then meet the other waves -- their LDS-DMA shares of the A tile are then in
    // place.  Bare s_barrier: __syncthreads() carries a fence that would drain the younger chunk as well.  Nothing ties
    // the weight registers to the wait (an in/out operand would make the compiler copy them BEFORE the wait, i.e. while
    // the load is in flight); instead every MFMA also consumes an LDS operand that a volatile asm after the wait produces,
    // and volatile asm statements keep their order.
    auto landed = [&](int younger_in_flight) {
        if (DEPTH > 2 && younger_in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * (8 + XV)) : "memory");
        else if (younger_in_flight >= 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(8 + XV) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // one phase: MFMAs of chunk c out of `cur` / LDS buffer `buf`, then the same registers are refilled with chunk c + DEPTH
    // and LDS buffer (buf - 1) mod NB, which every wave finished reading before this phase's barrier
    auto phase = [&](u32x4 (&cur)[4][2], int c, int buf) {
        landed(c1 - 1 - c);
        const uint32_t xbase = lds_base + (uint32_t)buf * (R * GEMM_KC * 2) + (uint32_t)n * (GEMM_KC * 2);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t u0 = (uint32_t)((8 * b + 2 * g) ^ n) * 16, u1 = (uint32_t)((8 * b + 2 * g + 1) ^ n) * 16;   // rows 16 mt + n: (row & 15) == n
            // all row tiles of this k block in one LDS round trip (row tile mt sits 16 rows = 8 KiB further: immediate offsets)
            const uint32_t a0 = xbase + u0, a1 = xbase + u1;
            u32x4 r[RT][2];
            if constexpr (RT == 1)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]) : "v"(a0), "v"(a1));
            else if constexpr (RT == 2)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:8192\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]) : "v"(a0), "v"(a1));
            else if constexpr (RT == 3)
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %6 offset:8192\n\tds_read_b128 %3, %7 offset:8192\n\t"
                             "ds_read_b128 %4, %6 offset:16384\n\tds_read_b128 %5, %7 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]) : "v"(a0), "v"(a1));
            else
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %9 offset:8192\n\t"
                             "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %9 offset:16384\n\tds_read_b128 %6, %8 offset:24576\n\t"
                             "ds_read_b128 %7, %9 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[3][0]), "=&v"(r[3][1])
                             : "v"(a0), "v"(a1));
#pragma unroll
            for (int mt = 0; mt < RT; mt++) {
                acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][0]), __builtin_bit_cast(V8, cur[b][0]), acc[mt]);
                acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][1]), __builtin_bit_cast(V8, cur[b][1]), acc[mt]);
            }
            // 64 rows: refill as soon as this k block's operands are consumed -- the loads trickle into the memory pipe between
            // the k blocks instead of arriving as one burst after a load-free compute window (all 8 waves compute at once):
            // gate|up 40.5 -> 39.8 us; neutral at 16 / 32 rows, which keep the burst (profiles/r02_gemm_rows64.md)
            if (RT >= 3 && c + DEPTH < c1) {
                load_wb(cur, c + DEPTH, b);
                if (b < XV) stage_xi(c + DEPTH, buf == 0 ? NB - 1 : buf - 1, b);
            }
        }
        if (RT < 3 && c + DEPTH < c1) { load_w(cur, c + DEPTH); stage_x(c + DEPTH, buf == 0 ? NB - 1 : buf - 1); }
    };
    if (c0 < c1) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (c0 + d < c1) { load_w(wr[d], c0 + d); stage_x(c0 + d, d); }
        int buf = 0;
        for (int c = c0; c < c1; c += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                if (c + d < c1) { phase(wr[d], c + d, buf); buf = buf == NB - 1 ? 0 : buf + 1; }
        }
    }
    // C layout of mfma_16x16: lane holds rows 4g + r of column n
    if constexpr (EPI == 1) {
        float *ex = reinterpret_cast<float *>(gemm_lds);            // [R][64] up values; the A tiles are dead by now
        __syncthreads();
        if (w >= 4) {
#pragma unroll
            for (int mt = 0; mt < RT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) ex[(16 * mt + 4 * g + r) * 64 + 16 * (w - 4) + n] = acc[mt][r];
        }
        __syncthreads();
        if (w < 4) {
#pragma unroll
            for (int mt = 0; mt < RT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = 16 * mt + 4 * g + r;
                    // the roundings of HF's act_fn(gate_proj(x)) * up_proj(x) in the model dtype (same as k_silu_mul)
                    const float gf = (float)(E)acc[mt][r], uf = (float)(E)ex[m * 64 + 16 * w + n];
                    const E sv = (E)(gf / (1.f + __expf(-gf)));
                    out[(size_t)m * (N / 2) + blockIdx.x * 64 + 16 * w + n] = (E)((float)sv * uf);
                }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < RT; mt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = 16 * mt + 4 * g + r;
            if (out) out[(size_t)m * N + n0 + n] = (E)acc[mt][r];
            // write-through (sc1) stores: the fp32 partials leave the L2 while the launch still streams, instead of as dirty lines the
            // kernel boundary has to flush (64 rows: 46 MB per layer; 3.93 -> 3.89 ms per step, neutral at 16 rows)
            else __hip_atomic_store(&partial[((size_t)split * R + m) * N + n0 + n], acc[mt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// RMSNorm applied by the CONSUMING projection (NORM variants of k_gemm_qkv_rope / k_gemm_pairs_silu): the rows' sums of squares arrive as
// per-16-column partials (k_gemm_cs_residual, k_embed_rows_ssq), every workgroup adds them up in the same fixed order and keeps
// 1 / rms per row in LDS; the A chunks then pass through registers instead of the LDS DMA and are scaled there with the reference's
// roundings: h = (x * rsqrt(mean(x^2) + eps)).to(dtype), a = weight * h (LlamaRMSNorm.forward).
struct NormArgs { const float *ssq; const void *g; int tiles; float inv_hidden; float eps; };

// the partial sums are REQUESTED before the first weight chunks (hand-issued loads, so that the counted waits of the stream stay exact) and
// added up once those chunks are in flight: up to NORM_NS loads per thread (rounds past the tile count are skipped, the last one is clamped + zeroed)
#define NORM_NS 16
template <int NT>
__device__ __forceinline__ void norm_issue(const NormArgs &na, float (&sv)[NORM_NS]) {
    const int tid = threadIdx.x, row = tid & 15, p = tid >> 4;
#pragma unroll
    for (int k = 0; k < NORM_NS; k++) {
        sv[k] = 0.f;
        if ((NT / 16) * k >= na.tiles) continue;             // (uniform) nothing of this round exists: K = 4096 needs 8 of the 16 rounds
        int t = p + (NT / 16) * k;
        t = t < na.tiles ? t : na.tiles - 1;
        const float *src = na.ssq + (size_t)t * 16 + row;
        asm volatile("global_load_dword %0, %1, off" : "=v"(sv[k]) : "v"(src) : "memory");
    }
}
template <int NT>
__device__ __forceinline__ void norm_finish(const NormArgs &na, float (&sv)[NORM_NS], float *part /* [NT / 16][16] */, float *rs /* [16] */) {
    const int tid = threadIdx.x, row = tid & 15, p = tid >> 4;
    // the caller's counted wait has just retired the loads of norm_issue.  The compiler does not know that those asm statements were
    // asynchronous: this statement re-defines their destinations HERE, so that no use can be scheduled above the wait and the registers
    // stay reserved until it (without it a build with branches around the loads consumed a register one instruction after its load was
    // issued and recycled it as an address while the load was in flight)
    static_assert(NORM_NS == 16, "");
    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]),
                      "+v"(sv[8]), "+v"(sv[9]), "+v"(sv[10]), "+v"(sv[11]), "+v"(sv[12]), "+v"(sv[13]), "+v"(sv[14]), "+v"(sv[15]) : : "memory");
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_NS; k++) s += (p + (NT / 16) * k < na.tiles) ? sv[k] : 0.f;
    part[p * 16 + row] = s;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < 16) {
        float tot = 0.f;
        for (int k = 0; k < NT / 16; k++) tot += part[k * 16 + tid];
        rs[tid] = rsqrtf(tot * na.inv_hidden + na.eps);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename E>
__device__ __forceinline__ u32x4 norm_scale8(u32x4 xraw, u32x4 graw, float rs) {
    const E *xe = reinterpret_cast<const E *>(&xraw);
    const E *ge = reinterpret_cast<const E *>(&graw);
    u32x4 out;
    E *oe = reinterpret_cast<E *>(&out);
#pragma unroll
    for (int j = 0; j < 8; j++) { const E h = (E)((float)xe[j] * rs); oe[j] = (E)((float)ge[j] * (float)h); }
    return out;
}


// counted wait of the weight stream: memory operations retire in issue order, so chunk c has landed when at most `younger` x PC
// operations (PC = this wave's memory operations per chunk) may still fly
template <int DEPTH, int PC>
__device__ __forceinline__ void gemm_wait_younger(int younger) {
    if constexpr (PC == 0) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    else {
        if (DEPTH > 3 && younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(3 * PC) : "memory");
        else if (DEPTH > 2 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PC) : "memory");
        else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }
}

// ================================================================================================
// q|k|v projection with RoPE and the K/V row write as its epilogue (round 3): the k_rope_kv launch and the fp32 split-K partials of the
// q|k|v projection disappear (Vicuna-7B: 5 us and 1.5 MB written + read back per layer).  An epilogue needs COMPLETE sums, so no split-K
// across workgroups: the launch gets its workgroups from 64-column tiles instead -- (H + 2 H_kv) x 2 of them, 192 for a 32-head MHA
// model -- and the two halves of every 256-k chunk go to two wave groups of the same workgroup (waves 0-3: k blocks 0,1; waves 4-7:
// k blocks 2,3; 16 columns each), whose accumulators meet in LDS after the stream.  A tile holds 32 complete rotate_half PAIRS of one
// head -- head columns {32 half + i, 64 + 32 half + i : i < 32} -- so the rotation needs nothing from another workgroup; the row
// permutation is applied once, when the weights are packed (samd_gemm_pack_qkv64).
//   PACKED LAYOUT (64-column tiles): block (tile t, chunk c) = 32 KiB contiguous at ((t * K/256 + c) * 2048) uint4 units; unit
//   (2 bl + j) * 512 + tid holds Wperm[64 t + 16 cg + n][256 c + 64 (2 kh + bl) + 16 g + 8 j .. +7] for tid = 64 (4 kh + cg) + 16 g + n.
// The stream itself is k_gemm_skinny's: LDS-DMA'd A chunks shared by all waves, hand-issued nt weight loads with DEPTH chunks in flight
// (32 KiB chunks: DEPTH 4 = the same 128 KiB per workgroup), counted waits, bare barriers.
// ================================================================================================
// AR (NORM only; round 4): activation rows a workgroup fetches -- 16, or 8 for a draft of <= 8 nodes: waves 0-3 stage rows 0-7, waves 4-7
// stage nothing (rows 8-15 of the LDS tiles are zeroed once), a sixth less of what the CU's memory pipe ingests beside the weights
template <typename TT, int RT, int DEPTH, int CG, bool NORM, int AR = 16>
__global__ __launch_bounds__(64 * GEMM_WAVES, RT >= 3 || NORM ? 2 : GEMM_WAVES / 2) void k_gemm_qkv_rope(
        const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ W, int K, int n_chunks,
        const float *__restrict__ cs, const int *__restrict__ d_L, const int *__restrict__ d_n,
        typename TT::elem *__restrict__ q_out, typename TT::elem *__restrict__ k_cache, typename TT::elem *__restrict__ v_cache,
        int H, int Hkv, long long max_len, NormArgs na, int v_t) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    constexpr int R = 16 * RT;
    constexpr int NT = 64 * GEMM_WAVES;
    constexpr int XV = (R * 32) / NT;              // 16-byte units per thread to stage one A chunk
    constexpr int NB = DEPTH + 1;
    constexpr int WL = 4;                          // weight loads per lane and chunk (streaming waves)
    constexpr int TW = 16 * CG, PP = 8 * CG;       // tile width in columns, rotate_half pairs per tile
    constexpr int CH = CG * 8192;                  // bytes of one (tile, chunk) block
    extern __shared__ __attribute__((aligned(1024))) char gemm_lds[];
    E (*xs)[R][GEMM_KC] = reinterpret_cast<E (*)[R][GEMM_KC]>(gemm_lds);

    // waves 0 .. 2 CG - 1 stream and multiply (column group cg, k half kh); with 48-column tiles waves 6, 7 only help staging A
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const bool streams = CG == 4 || w < 2 * CG;
    static_assert(AR == 16 || (AR == 8 && NORM && RT == 1), "8 activation rows: the norm-fold kernels only");
    const bool stager = AR == 16 || w < GEMM_WAVES / 2;                          // wave-uniform: rows 0..7 belong to threads 0..255
    const int cg = CG == 4 ? (w & 3) : (streams ? w % CG : 0), kh = CG == 4 ? (w >> 2) : (streams ? w / CG : 0);
    const char *wtile = reinterpret_cast<const char *>(W) + (size_t)blockIdx.x * n_chunks * CH;
    const uint32_t wlane = (uint32_t)(64 * (CG * kh + cg) + l) * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)&xs[0][0][0];
    // the scalars and this thread's cos | sin are requested now and used after the stream
    const int n_rows = d_n[0], L = d_L[0];

    floatx4 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; mt++) acc[mt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    u32x4 wr[DEPTH][2][2];
    auto load_wb = [&](u32x4 (&dst)[2][2], int c, int bl) {
        const char *p = wtile + (size_t)c * CH;
#pragma unroll
        for (int j = 0; j < 2; j++)
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst[bl][j]) : "v"(wlane), "s"(p + (CH / 4) * (2 * bl + j)) : "memory");
    };
    // NORM: the A chunk and the norm weights of its k range come through registers (two loads per unit) and are scaled on their way to LDS
    static_assert(!NORM || RT == 1, "the norm-fold path is built for the 16-row tile");
    constexpr int XL = NORM ? 2 * XV : XV;         // memory operations per thread to stage one A chunk
    u32x4 xr[NORM ? DEPTH : 1][NORM ? XV : 1], gr[NORM ? DEPTH : 1][NORM ? XV : 1];
    __shared__ float norm_part[NORM ? (NT / 16) * 16 : 1], norm_rs[NORM ? 16 : 1];
    auto stage_xi = [&](int c, int buf, int i, int d) {
        const int slot = tid + NT * i, row = slot >> 5, pos = slot & 31, unit = pos ^ (row & 15);
        const E *src = A + (size_t)row * K + (size_t)c * GEMM_KC + 8 * unit;
        if constexpr (NORM) {
            const E *gsrc = reinterpret_cast<const E *>(na.g) + (size_t)c * GEMM_KC + 8 * unit;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[d][i]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gr[d][i]) : "v"(gsrc) : "memory");
        } else {
            E *dst = &xs[buf][0][0] + (size_t)(NT * i + 64 * w) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            asm volatile("" ::: "memory");
        }
    };
    auto issue = [&](u32x4 (&dst)[2][2], int c, int buf, int d) {
        if (streams) { load_wb(dst, c, 0); load_wb(dst, c, 1); }
        if (stager) {
#pragma unroll
            for (int i = 0; i < XV; i++) stage_xi(c, buf, i, d);
        }
    };
    auto scale_to_lds = [&](int buf, int d) {      // NORM: this thread's units of the landed chunk -> LDS, scaled (same positions as the DMA's)
        if constexpr (NORM) {
            if (!stager) return;
#pragma unroll
            for (int i = 0; i < XV; i++) {
                const int slot = tid + NT * i, row = slot >> 5;
                // (re-defined here, behind the counted wait: the compiler does not know that the asm loads that produced them were asynchronous)
                asm volatile("" : "+v"(xr[d][i]), "+v"(gr[d][i]) : : "memory");
                const u32x4 v = norm_scale8<E>(xr[d][i], gr[d][i], norm_rs[row]);
                *reinterpret_cast<u32x4 *>(&xs[buf][0][0] + (size_t)slot * 8) = v;
            }
        }
    };
    auto landed = [&](int younger, int buf, int d) {   // memory ops retire in issue order: chunk c has landed when only the younger ones may still fly
        if (streams) { if (stager) gemm_wait_younger<DEPTH, WL + XL>(younger); else gemm_wait_younger<DEPTH, WL>(younger); }
        else { if (stager) gemm_wait_younger<DEPTH, XL>(younger); else gemm_wait_younger<DEPTH, 0>(younger); }   // (a staging-only / an idle wave)
        if constexpr (NORM) { scale_to_lds(buf, d); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto phase = [&](u32x4 (&cur)[2][2], int c, int buf, int d) {
        landed(n_chunks - 1 - c, buf, d);
        const uint32_t xbase = lds_base + (uint32_t)buf * (R * GEMM_KC * 2) + (uint32_t)n * (GEMM_KC * 2);
#pragma unroll
        for (int bl = 0; bl < 2; bl++) {
            if (!streams) break;
            const int b = 2 * kh + bl;
            const uint32_t a0 = xbase + (uint32_t)((8 * b + 2 * g) ^ n) * 16, a1 = xbase + (uint32_t)((8 * b + 2 * g + 1) ^ n) * 16;
            u32x4 r[RT][2];
            if constexpr (RT == 1)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]) : "v"(a0), "v"(a1));
            else if constexpr (RT == 2)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:8192\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]) : "v"(a0), "v"(a1));
            else if constexpr (RT == 3)
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %6 offset:8192\n\tds_read_b128 %3, %7 offset:8192\n\t"
                             "ds_read_b128 %4, %6 offset:16384\n\tds_read_b128 %5, %7 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]) : "v"(a0), "v"(a1));
            else
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %9 offset:8192\n\t"
                             "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %9 offset:16384\n\tds_read_b128 %6, %8 offset:24576\n\t"
                             "ds_read_b128 %7, %9 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[3][0]), "=&v"(r[3][1])
                             : "v"(a0), "v"(a1));
#pragma unroll
            for (int mt = 0; mt < RT; mt++) {
                acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][0]), __builtin_bit_cast(V8, cur[bl][0]), acc[mt]);
                acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][1]), __builtin_bit_cast(V8, cur[bl][1]), acc[mt]);
            }
        }
        if (c + DEPTH < n_chunks) issue(cur, c + DEPTH, buf == 0 ? NB - 1 : buf - 1, d);
    };
    float norm_sv[NORM ? NORM_NS : 1];
    if constexpr (NORM) norm_issue<NT>(na, norm_sv);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < n_chunks) issue(wr[d], d, d, d);
    if constexpr (AR == 8) {                        // rows 8..15 of every A buffer: zero, once (the waves that stage nothing write them)
        if (!stager) {
#pragma unroll
            for (int bz = 0; bz < NB; bz++) *reinterpret_cast<u32x4 *>(&xs[bz][0][0] + (size_t)tid * 8) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    if constexpr (NORM) {                           // the partial sums were requested before the chunks: they have landed when only the chunks' operations fly
        if (n_chunks >= DEPTH) {
            if (streams) { if (stager) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * (WL + XL)) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * WL) : "memory"); }
            else { if (stager) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * XL) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
        } else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        norm_finish<NT>(na, norm_sv, norm_part, norm_rs);
    }
    {
        int buf = 0;
        for (int c = 0; c < n_chunks; c += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                if (c + d < n_chunks) { phase(wr[d], c + d, buf, d); buf = buf == NB - 1 ? 0 : buf + 1; }
        }
    }
    // ---- epilogue: the two k halves meet in LDS ([2][R][TW] fp32; the A tiles are dead), then RoPE + rounding + row writes -------------
    float *ex = reinterpret_cast<float *>(gemm_lds);
    __syncthreads();
    if (streams) {
#pragma unroll
        for (int mt = 0; mt < RT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) ex[(kh * R + 16 * mt + 4 * g + r) * TW + 16 * cg + n] = acc[mt][r];     // C layout: lane holds rows 4g + r of column n
    }
    __syncthreads();
    // v_t (round 6): the V cache is transposed ([H_kv][128][max_len], samd_tree_attention_vt); a tile of V columns then walks its (row, pair)
    // items row-fastest, so that neighbouring lanes write neighbouring keys of one V^T row
    const bool vt_tile = v_t && ((PP * (int)blockIdx.x) >> 6) >= H + Hkv;
    for (int i = tid; i < R * PP; i += NT) {
        const int row = vt_tile ? i % R : i / PP, p = vt_tile ? i / R : i % PP;
        if (row >= n_rows || L + row >= max_len) continue;                     // rows past the draft / past the cache are not written (k_rope_kv)
        // sums rounded to the model dtype first, as the projection's own output would have been (k_rope_kv does the same on partials)
        const float x1 = (float)(E)(ex[row * TW + p] + ex[(R + row) * TW + p]);
        const float x2 = (float)(E)(ex[row * TW + PP + p] + ex[(R + row) * TW + PP + p]);
        const int pair = PP * (int)blockIdx.x + p, head = pair >> 6, j = pair & 63;   // global pair index -> head, position inside it (j and j + 64)
        if (head >= H + Hkv) {                                                    // V: plain rows
            if (v_t) {
                E *dst = v_cache + (size_t)(head - H - Hkv) * max_len * 128 + L + row;
                dst[(size_t)j * max_len] = (E)x1; dst[(size_t)(j + 64) * max_len] = (E)x2;
                continue;
            }
            E *dst = v_cache + ((size_t)(head - H - Hkv) * max_len + L + row) * 128;
            dst[j] = (E)x1; dst[j + 64] = (E)x2;
            continue;
        }
        const float c = cs[(size_t)row * 128 + j], sn = cs[(size_t)row * 128 + 64 + j];
        const E o1 = (E)(x1 * c - x2 * sn), o2 = (E)(x2 * c + x1 * sn);
        E *dst = head < H ? q_out + ((size_t)row * H + head) * 128 : k_cache + ((size_t)(head - H) * max_len + L + row) * 128;
        dst[j] = o1; dst[j + 64] = o2;
    }
}

// row-major [N][K] -> the packed tile layout of k_gemm_qkv_rope<CG> (16 CG columns per tile), with the rotate_half row permutation:
// pairs are numbered head by head (pair P = 64 head + j stands for head columns j and 64 + j); tile t holds pairs [8 CG t, 8 CG (t + 1)) --
// packed row 16 CG t + q = the FIRST column of pair 8 CG t + q for q < 8 CG, the SECOND column of pair 8 CG t + q - 8 CG otherwise.
//   block (tile t, chunk c) = CG x 8 KiB contiguous at ((t * K/256 + c) * 512 CG) uint4 units; unit (2 bl + j) * 128 CG + x holds
//   Wperm[16 CG t + 16 cg + n][256 c + 64 (2 kh + bl) + 16 g + 8 j .. +7] for x = 64 (CG kh + cg) + 16 g + n.
template <int CG>
__global__ __launch_bounds__(256) void k_gemm_pack_qkv(const uint4 *__restrict__ W, uint4 *__restrict__ out, int N, int K) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * K / 8;
    if (u >= total) return;
    constexpr int PP = 8 * CG;
    const int n_chunks = K / GEMM_KC;
    const long long blk = u / (512 * CG);
    const int in = (int)(u % (512 * CG)), jj = in / (128 * CG), x = in % (128 * CG), w = x >> 6, g = (x >> 4) & 3, n = x & 15, cg = w % CG, kh = w / CG;
    const int t = (int)(blk / n_chunks), c = (int)(blk % n_chunks), bl = jj >> 1, j = jj & 1;
    const int q = 16 * cg + n, pair = PP * t + (q < PP ? q : q - PP);
    const long long row = 128LL * (pair >> 6) + (pair & 63) + (q < PP ? 0 : 64), col = 256LL * c + 64 * (2 * kh + bl) + 16 * g + 8 * j;
    out[u] = W[(row * K + col) / 8];
}

// 48-column tiles (3 column groups) when they divide the matrix and need no more rounds of workgroups over the CUs than 64-column tiles:
// Vicuna-7B's 12288 q|k|v columns = 256 tiles of 48 = one workgroup on every CU (64-column tiles: 192 workgroups, a quarter of the CUs idle
// in a launch bound by what a CU's memory pipe ingests); Vicuna-13B's 15360 = 320 x 48 (two rounds) or 240 x 64 (one): 64.
static int qkv_tile_groups(int n_heads_total) {
    static const int env = [] { const char *e = getenv("SAMD_QKV_TILE"); return e ? atoi(e) : 0; }();
    const int n_cu = samd_cu_count();
    const int N = n_heads_total * 128;
    if (N % 48 != 0 || env == 64) return 4;
    if (env == 48) return 3;
    const int t48 = N / 48, t64 = N / 64;
    return ((t48 + n_cu - 1) / n_cu) * 48 < ((t64 + n_cu - 1) / n_cu) * 64 ? 3 : 4;
}

// ================================================================================================
// gate|up projection + SiLU*up on ALL CUs (round 3).  k_gemm_skinny<EPI = 1> covers gate|up with 128-column tiles: 172 workgroups for
// Vicuna-7B's 2 x 11008 columns, and a launch on 172 CUs streams at 172 x ~36 GB/s = 6.0 TB/s -- what a CU's memory pipe ingests,
// not what HBM delivers (profiles/r03_gemm_variants.md).  Here the unit of work is a PAIR = 16 gate columns + the 16 up columns they
// multiply (two MFMA column groups, two waves): 688 pairs are dealt out evenly, 2 or 3 to each of 256 workgroups, so every CU pulls.
// Waves 2i / 2i + 1 of a workgroup own pair i's gate / up group; waves beyond the workgroup's share only help staging A.
//   PACKED LAYOUT (group-major): column group gi (16 columns), chunk c: 8 KiB contiguous at ((gi * K/256 + c) * 512) uint4 units;
//   unit (2b + j) * 64 + lane holds Wg[16 gi + n][256 c + 64 b + 16 g + 8 j .. +7] for lane = 16 g + n, where Wg = the gate|up matrix with
//   its rows interleaved in groups of 16 (group 2p = gate rows 16p.., group 2p + 1 = up rows 16p..) -- samd_gemm_pack_groups.
// Stream, A staging and waits are k_gemm_skinny's (two chunks in flight, counted vmcnt, bare barriers).
// ================================================================================================
template <typename TT, int RT, int DEPTH, bool NORM, int AR = 16>          // AR: see k_gemm_qkv_rope
__global__ __launch_bounds__(64 * GEMM_WAVES, RT >= 3 || DEPTH > 2 ? 2 : GEMM_WAVES / 2) void k_gemm_pairs_silu(const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ W,
                                                                                                     typename TT::elem *__restrict__ out, int K, int inter, int n_chunks, int n_pairs, NormArgs na) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    constexpr int R = 16 * RT;
    constexpr int NT = 64 * GEMM_WAVES;
    constexpr int XV = (R * 32) / NT;
    constexpr int NB = DEPTH + 1;
    extern __shared__ __attribute__((aligned(1024))) char gemm_lds[];
    E (*xs)[R][GEMM_KC] = reinterpret_cast<E (*)[R][GEMM_KC]>(gemm_lds);

    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, g = l >> 4;
#ifdef SAMD_GEMM_ABLATE
    const int abl = samd_abl_flag;
#else
    constexpr int abl = 0;
#endif
    // this workgroup's pairs [p0, p1): an even deal of n_pairs over the grid
    const int p0 = (int)((long long)blockIdx.x * n_pairs / gridDim.x), p1 = (int)((long long)(blockIdx.x + 1) * n_pairs / gridDim.x);
    const bool active = w < 2 * (p1 - p0);                                       // wave-uniform
    static_assert(AR == 16 || (AR == 8 && NORM && RT == 1), "8 activation rows: the norm-fold kernels only");
    const bool stager = AR == 16 || w < GEMM_WAVES / 2;                          // wave-uniform: rows 0..7 belong to threads 0..255
    const int gi = 2 * p0 + w;                                                    // this wave's column group
    const char *wgrp = reinterpret_cast<const char *>(W) + (size_t)(active ? gi : 0) * n_chunks * 8192;
    const uint32_t wlane = (uint32_t)l * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)&xs[0][0][0];

    floatx4 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; mt++) acc[mt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    u32x4 wr[DEPTH][4][2];
    auto load_w = [&](u32x4 (&dst)[4][2], int c) {
        const char *p = wgrp + (size_t)c * 8192;
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst[b][j]) : "v"(wlane), "s"(p + 1024 * (2 * b + j)) : "memory");
    };
    // NORM: the A chunk and the norm weights of its k range come through registers and are scaled on their way to LDS (see NormArgs)
    static_assert(!NORM || RT == 1, "the norm-fold path is built for the 16-row tile");
    constexpr int XL = NORM ? 2 * XV : XV;         // memory operations per thread to stage one A chunk
    u32x4 xr[NORM ? DEPTH : 1][NORM ? XV : 1], gr[NORM ? DEPTH : 1][NORM ? XV : 1];
    __shared__ float norm_part[NORM ? (NT / 16) * 16 : 1], norm_rs[NORM ? 16 : 1];
    auto stage_x = [&](int c, int buf, int d) {
        if (!stager || (abl & 4)) return;
#pragma unroll
        for (int i = 0; i < XV; i++) {
            const int slot = tid + NT * i, row = slot >> 5, pos = slot & 31, unit = pos ^ (row & 15);
            const E *src = A + (size_t)((abl & 8) ? (row & 15) : row) * K + (size_t)c * GEMM_KC + 8 * unit;
            if constexpr (NORM) {
                const E *gsrc = reinterpret_cast<const E *>(na.g) + (size_t)c * GEMM_KC + 8 * unit;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[d][i]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gr[d][i]) : "v"(gsrc) : "memory");
            } else {
                E *dst = &xs[buf][0][0] + (size_t)(NT * i + 64 * w) * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
                asm volatile("" ::: "memory");
            }
        }
    };
    // a wave without a column group issues no weight loads, one that stages no rows no A loads: its counted waits leave those out
    auto landed = [&](int younger, int buf, int d) {
        const bool stg = stager && !(abl & 4);
        if (active) { if (stg) gemm_wait_younger<DEPTH, 8 + XL>(younger); else gemm_wait_younger<DEPTH, 8>(younger); }
        else { if (stg) gemm_wait_younger<DEPTH, XL>(younger); else gemm_wait_younger<DEPTH, 0>(younger); }
        if constexpr (NORM) {
            if (stager) {
#pragma unroll
                for (int i = 0; i < XV; i++) {
                    const int slot = tid + NT * i, row = slot >> 5;
                    // (re-defined here, behind the counted wait: the compiler does not know that the asm loads that produced them were asynchronous)
                    asm volatile("" : "+v"(xr[d][i]), "+v"(gr[d][i]) : : "memory");
                    const u32x4 v = norm_scale8<E>(xr[d][i], gr[d][i], norm_rs[row]);
                    *reinterpret_cast<u32x4 *>(&xs[buf][0][0] + (size_t)slot * 8) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (!(abl & 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto phase = [&](u32x4 (&cur)[4][2], int c, int buf, int d) {
        landed(n_chunks - 1 - c, buf, d);
        if (active && !(abl & 2)) {
            const uint32_t xbase = lds_base + (uint32_t)buf * (R * GEMM_KC * 2) + (uint32_t)n * (GEMM_KC * 2);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t a0 = xbase + (uint32_t)((8 * b + 2 * g) ^ n) * 16, a1 = xbase + (uint32_t)((8 * b + 2 * g + 1) ^ n) * 16;
                u32x4 r[RT][2];
                if constexpr (RT == 1)
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]) : "v"(a0), "v"(a1));
                else if constexpr (RT == 2)
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:8192\n\t"
                                 "s_waitcnt lgkmcnt(0)" : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]) : "v"(a0), "v"(a1));
                else if constexpr (RT == 3)
                    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %6 offset:8192\n\tds_read_b128 %3, %7 offset:8192\n\t"
                                 "ds_read_b128 %4, %6 offset:16384\n\tds_read_b128 %5, %7 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]) : "v"(a0), "v"(a1));
                else
                    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %9 offset:8192\n\t"
                                 "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %9 offset:16384\n\tds_read_b128 %6, %8 offset:24576\n\t"
                                 "ds_read_b128 %7, %9 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[3][0]), "=&v"(r[3][1])
                                 : "v"(a0), "v"(a1));
#pragma unroll
                for (int mt = 0; mt < RT; mt++) {
                    acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][0]), __builtin_bit_cast(V8, cur[b][0]), acc[mt]);
                    acc[mt] = TT::mfma(__builtin_bit_cast(V8, r[mt][1]), __builtin_bit_cast(V8, cur[b][1]), acc[mt]);
                }
            }
        }
        if (c + DEPTH < n_chunks) { if (active) load_w(cur, c + DEPTH); stage_x(c + DEPTH, buf == 0 ? NB - 1 : buf - 1, d); }
    };
    float norm_sv[NORM ? NORM_NS : 1];
    if constexpr (NORM) norm_issue<NT>(na, norm_sv);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < n_chunks) { if (active) load_w(wr[d], d); stage_x(d, d, d); }
    if constexpr (AR == 8) {                        // rows 8..15 of every A buffer: zero, once
        if (!stager) {
#pragma unroll
            for (int bz = 0; bz < NB; bz++) *reinterpret_cast<u32x4 *>(&xs[bz][0][0] + (size_t)tid * 8) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    if constexpr (NORM) {                           // the partial sums were requested before the chunks (see k_gemm_qkv_rope)
        if (n_chunks >= DEPTH) {
            if (active) { if (stager) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * (8 + XL)) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * 8) : "memory"); }
            else { if (stager) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH * XL) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
        } else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        norm_finish<NT>(na, norm_sv, norm_part, norm_rs);
    }
    {
        int buf = 0;
        for (int c = 0; c < n_chunks; c += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                if (c + d < n_chunks) { phase(wr[d], c + d, buf, d); buf = buf == NB - 1 ? 0 : buf + 1; }
        }
    }
    // ---- epilogue: the up waves hand their values over through LDS, the gate waves write silu(gate) * up ---------------------------------
    float *ex = reinterpret_cast<float *>(gemm_lds);                            // [4 pairs][R][16]; the A tiles are dead
    __syncthreads();
    if (active && (w & 1)) {
#pragma unroll
        for (int mt = 0; mt < RT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) ex[((w >> 1) * R + 16 * mt + 4 * g + r) * 16 + n] = acc[mt][r];
    }
    __syncthreads();
    if (active && !(w & 1)) {
        const int col = 16 * (p0 + (w >> 1)) + n;
#pragma unroll
        for (int mt = 0; mt < RT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = 16 * mt + 4 * g + r;
                // the roundings of HF's act_fn(gate_proj(x)) * up_proj(x) in the model dtype (same as k_silu_mul)
                const float gf = (float)(E)acc[mt][r], uf = (float)(E)ex[((w >> 1) * R + m) * 16 + n];
                const E sv = (E)(gf / (1.f + __expf(-gf)));
                out[(size_t)m * inter + col] = (E)((float)sv * uf);
            }
    }
}

// ================================================================================================
// Complete-sum projection with the residual add as its epilogue (round 3, the "norm-fold" forward): x[m][n] <- x[m][n] + (A W^T)[m][n] for
// o_proj / down_proj at 16 rows, plus the row's sum of squares over the workgroup's 16 columns -- what the CONSUMING projection needs to
// apply the RMSNorm itself (k_gemm_qkv_rope / k_gemm_pairs_silu with NORM), so that the two k_rmsnorm launches of a decoder layer and the
// fp32 split-K partials they read disappear.  One workgroup per 16 output columns and ALL of K: wave w takes the 256-k chunks w, w + 8, ...
// with its own LDS-DMA'd A chunks (no barrier in the stream), the eight partial accumulators meet in LDS in a fixed order.
//   W: group-major packed (samd_gemm_pack_groups of the plain [N][K] matrix); ssq: [N / 16][16] fp32 (tile-major, row m at [t][m]).
// Roundings are the reference's: the projection's output in the model dtype, then the residual sum in the model dtype
// (LlamaDecoderLayer: hidden_states = residual + hidden_states), squares of the stored values (LlamaRMSNorm: x.float().pow(2)).
// ================================================================================================
// EARLY (seam experiment, scripts/seam_probe.py / profiles/r04_attention.md section 5; K = 4096 only: two chunks per wave): the launch runs
// on a second queue BESIDE the attention launches that produce A.  Every wave requests ALL of its weights at entry, then the workgroup
// polls `counter` -- the producers' arrival count -- until this workgroup's own epoch says A is complete, and only then requests A.
template <typename TT, int ROWS, bool EARLY = false>
__global__ __launch_bounds__(64 * GEMM_WAVES, 2) void k_gemm_cs_residual(const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ W,
                                                                           typename TT::elem *__restrict__ x, float *__restrict__ ssq, int K, int N, int n_chunks,
                                                                           const int *__restrict__ counter = nullptr, int *__restrict__ epoch = nullptr,
                                                                           int arrivals = 0) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    constexpr int CSD = 2;                                     // chunks in flight per wave
    extern __shared__ __attribute__((aligned(1024))) char gemm_lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, g = l >> 4;
    E *xs = reinterpret_cast<E *>(gemm_lds) + (size_t)w * CSD * 16 * GEMM_KC;          // this wave's CSD buffers of [16][256]
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)xs;
    const char *wgrp = reinterpret_cast<const char *>(W) + (size_t)blockIdx.x * n_chunks * 8192;
    const uint32_t wlane = (uint32_t)l * 16;

    // ROWS = 8: a draft of <= 8 rows -- only rows 0..7 of the activation tile are fetched (half of what this ingest-bound launch pulls
    // from L2 besides its weights); rows 8..15 of the LDS tiles are zeroed once and their outputs are not written
    static_assert(ROWS == 8 || ROWS == 16, "");
    constexpr int AL = ROWS / 2;                               // A loads per lane and chunk
    if constexpr (ROWS == 8) {
#pragma unroll
        for (int buf = 0; buf < CSD; buf++)
#pragma unroll
            for (int i = 4; i < 8; i++)
                *reinterpret_cast<u32x4 *>(xs + (size_t)buf * 16 * GEMM_KC + (size_t)(64 * i + l) * 8) = (u32x4){0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
    u32x4 wr[CSD][4][2];
    auto issue_w = [&](u32x4 (&dst)[4][2], int c) {
        const char *p = wgrp + (size_t)c * 8192;
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst[b][j]) : "v"(wlane), "s"(p + 1024 * (2 * b + j)) : "memory");
    };
    auto issue_a = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < AL; i++) {                         // the wave's own A chunk: 32 units of 16 B per row, unit u of row r at position u ^ (r & 15)
            const int slot = l + 64 * i, row = slot >> 5, pos = slot & 31, unit = pos ^ (row & 15);
            const E *src = A + (size_t)row * K + (size_t)c * GEMM_KC + 8 * unit;
            E *dst_l = xs + (size_t)buf * 16 * GEMM_KC + (size_t)(64 * i) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst_l, 16, 0, 0);
            asm volatile("" ::: "memory");
        }
    };
    auto issue = [&](u32x4 (&dst)[4][2], int c, int buf) { issue_w(dst, c); issue_a(c, buf); };
    auto phase = [&](u32x4 (&cur)[4][2], int c, int buf, bool more) {
        // EARLY: both chunks' weights are older than any A load, so what may still fly while chunk 0 is consumed is chunk 1's A only
        if (more) { if constexpr (EARLY) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(AL) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(8 + AL) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t xbase = lds_base + (uint32_t)buf * (16 * GEMM_KC * 2) + (uint32_t)n * (GEMM_KC * 2);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t a0 = xbase + (uint32_t)((8 * b + 2 * g) ^ n) * 16, a1 = xbase + (uint32_t)((8 * b + 2 * g + 1) ^ n) * 16;
            u32x4 r0, r1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1));
            acc = TT::mfma(__builtin_bit_cast(V8, r0), __builtin_bit_cast(V8, cur[b][0]), acc);
            acc = TT::mfma(__builtin_bit_cast(V8, r1), __builtin_bit_cast(V8, cur[b][1]), acc);
        }
    };
    // chunks of this wave: w, w + 8, ...; two in flight, the LDS buffer of a chunk is refilled only after its reads (lgkmcnt(0) above)
    const int mine = (n_chunks - w + GEMM_WAVES - 1) / GEMM_WAVES;
    if constexpr (EARLY) {
        if (mine > 0) issue_w(wr[0], w);
        if (mine > 1) issue_w(wr[1], w + GEMM_WAVES);
        __shared__ int my_epoch;
        if (tid == 0) {
            const int e = epoch[blockIdx.x];
            const int target = (e + 1) * arrivals;
            for (int spin = 0; spin < (1 << 22); spin++) {                 // bounded: a lost producer ends in wrong sums, not in a hung GPU
                if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(4);
            }
            my_epoch = e;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (mine > 0) issue_a(w, 0);
        if (mine > 1) issue_a(w + GEMM_WAVES, 1);
        if (mine > 0) phase(wr[0], w, 0, mine > 1);
        if (mine > 1) phase(wr[1], w + GEMM_WAVES, 1, false);
        if (tid == 0) epoch[blockIdx.x] = my_epoch + 1;
    } else {
        if (mine > 0) issue(wr[0], w, 0);
        if (mine > 1) issue(wr[1], w + GEMM_WAVES, 1);
        for (int i = 0; i < mine; i += 2) {
            phase(wr[0], w + GEMM_WAVES * i, 0, i + 1 < mine);
            if (i + 2 < mine) issue(wr[0], w + GEMM_WAVES * (i + 2), 0);
            if (i + 1 < mine) {
                phase(wr[1], w + GEMM_WAVES * (i + 1), 1, i + 2 < mine);
                if (i + 3 < mine) issue(wr[1], w + GEMM_WAVES * (i + 3), 1);
            }
        }
    }
    // ---- the eight k shares meet in LDS (the A buffers are dead), summed in wave order; residual, rounding, row sums of squares -----------
    __syncthreads();
    float *ex = reinterpret_cast<float *>(gemm_lds);                               // [8][16][16]
#pragma unroll
    for (int r = 0; r < 4; r++) ex[(w * 16 + 4 * g + r) * 16 + n] = acc[r];        // C layout: lane holds rows 4g + r of column n
    __syncthreads();
    if (tid < 16 * ROWS) {
        const int row = tid >> 4, col = tid & 15;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < GEMM_WAVES; k++) sum += ex[(k * 16 + row) * 16 + col];
        E *xp = x + (size_t)row * N + 16 * blockIdx.x + col;
        const E o = (E)sum;
        const E y = (E)((float)*xp + (float)o);
        *xp = y;
        float q = (float)y * (float)y;
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
        if (col == 0) ssq[(size_t)blockIdx.x * 16 + row] = q;
    }
}

// row-major [N][K] -> group-major packed layout of k_gemm_pairs_silu (rows already in group order); one thread moves one 16-byte unit
__global__ __launch_bounds__(256) void k_gemm_pack_groups(const uint4 *__restrict__ W, uint4 *__restrict__ out, int N, int K) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * K / 8;
    if (u >= total) return;
    const int n_chunks = K / GEMM_KC;
    const long long blk = u >> 9;                                    // 512 units per 8 KiB block
    const int in = (int)(u & 511), jj = in >> 6, lane = in & 63, g = lane >> 4, n = lane & 15, b = jj >> 1, j = jj & 1;
    const long long gi = blk / n_chunks; const int c = (int)(blk % n_chunks);
    const long long row = 16 * gi + n, col = 256LL * c + 64 * b + 16 * g + 8 * j;
    out[u] = W[(row * K + col) / 8];
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

// DEPTH = 2 chunks in flight at every row tile.  Three (128 KiB of LDS at 64 rows) measured SLOWER: gate|up 40.3 vs 38.8 us at 64
// rows, 34.9 vs 34.5 at 32 -- the launch is not short of requests in flight (profiles/r02_gemm_rows64.md).
template <typename TT, int RT, int EPI, int DEPTH, bool GM = false>
static hipError_t gemm_launch(dim3 grid, hipStream_t st, const void *A, const void *W, float *partial, void *out, int K, int N, int chunks, int splits) {
    constexpr int lds = (DEPTH + 1) * 16 * RT * GEMM_KC * 2;
    if constexpr (lds > 65536) {
        static unsigned long long done = 0ull;                     // per-device (samd_common.h)
        const hipError_t attr = samd_reserve_lds((const void *)k_gemm_skinny<TT, RT, EPI, DEPTH, GM>, lds, &done);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((k_gemm_skinny<TT, RT, EPI, DEPTH, GM>), grid, dim3(64 * GEMM_WAVES), lds, st, (const typename TT::elem *)A, (const typename TT::elem *)W, partial,
                       (typename TT::elem *)out, K, N, chunks, splits);
    return hipSuccess;
}

template <int EPI, bool GM = false>
static hipError_t gemm_dispatch(int dtype, int rows_pad, dim3 grid, hipStream_t st, const void *A, const void *W, float *partial, void *out, int K, int N, int chunks,
                                int splits) {
#define GO(TT, RT, D) return gemm_launch<TT, RT, EPI, D, GM>(grid, st, A, W, partial, out, K, N, chunks, splits)
#define ROWS(TT) do { if (rows_pad == 16) GO(TT, 1, 2); else if (rows_pad == 32) GO(TT, 2, 2); else if (rows_pad == 48) GO(TT, 3, 2); else GO(TT, 4, 2); } while (0)
    if (dtype == SAMD_F16) ROWS(GF16); else ROWS(GBF16);
#undef ROWS
#undef GO
}

template <typename TT, int RT, int DEPTH, int CG, bool NORM = false, int AR = 16>
static hipError_t qkv_rope_launch(hipStream_t st, const void *A, const void *W, int K, int tiles, const float *cs, const int *d_L, const int *d_n, void *q, void *k, void *v,
                                  int H, int Hkv, long long max_len, int v_t, NormArgs na = NormArgs{nullptr, nullptr, 0, 0.f, 0.f}) {
    constexpr int lds_a = (DEPTH + 1) * 16 * RT * GEMM_KC * 2, lds_e = 2 * 16 * RT * 16 * CG * 4, lds = lds_a > lds_e ? lds_a : lds_e;
    if constexpr (lds > 60000) {
        static unsigned long long done = 0ull;
        const hipError_t attr = samd_reserve_lds((const void *)k_gemm_qkv_rope<TT, RT, DEPTH, CG, NORM, AR>, lds + 4096, &done);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((k_gemm_qkv_rope<TT, RT, DEPTH, CG, NORM, AR>), dim3(tiles), dim3(64 * GEMM_WAVES), lds, st, (const typename TT::elem *)A, (const typename TT::elem *)W, K, K / GEMM_KC,
                       cs, d_L, d_n, (typename TT::elem *)q, (typename TT::elem *)k, (typename TT::elem *)v, H, Hkv, max_len, na, v_t);
    return hipSuccess;
}

template <typename TT, int RT, int DEPTH, bool NORM = false, int AR = 16>
static hipError_t pairs_silu_launch(hipStream_t st, int grid, const void *A, const void *W, void *out, int K, int inter, int n_pairs,
                                    NormArgs na = NormArgs{nullptr, nullptr, 0, 0.f, 0.f}) {
    constexpr int lds_a = (DEPTH + 1) * 16 * RT * GEMM_KC * 2, lds_e = 4 * 16 * RT * 16 * 4, lds = lds_a > lds_e ? lds_a : lds_e;
    if constexpr (lds > 60000) {
        static unsigned long long done = 0ull;
        const hipError_t attr = samd_reserve_lds((const void *)k_gemm_pairs_silu<TT, RT, DEPTH, NORM, AR>, lds + 4096, &done);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((k_gemm_pairs_silu<TT, RT, DEPTH, NORM, AR>), dim3(grid), dim3(64 * GEMM_WAVES), lds, st, (const typename TT::elem *)A, (const typename TT::elem *)W,
                       (typename TT::elem *)out, K, inter, K / GEMM_KC, n_pairs, na);
    return hipSuccess;
}

extern "C" {

// choose the split-K factor.  Measured (scripts/gemm_bench.py, profiles/): the stream is fastest when the launch is ONE
// balanced wave of workgroups -- at most one per CU, each with a long run of chunks (gate|up: 172 workgroups x 16 chunks
// 5.47 TB/s, lm_head 250 x 16 5.92 TB/s; QKV 96 x 2 splits 5.15 TB/s vs 4.9 with 4 or 8) -- so: the largest split count
// that keeps columns x splits <= 256 and leaves every split at least one chunk.  Fewer splits also mean fewer fp32
// partials for the consumer to add up; capped at 8 (whole forward at 64 rows: 4.47 ms with cap 8, 4.56 with 6, 4.64 with 4).
// The two tuning knobs (SAMD_GEMM_SPLIT_CAP, SAMD_GEMM_SPLITS) are read ONCE, at the first call: callers size their fp32
// partial-sum workspaces from this function's answer, so the answer for a shape must not change during the process.
int samd_gemm_splits(int32_t N, int32_t K, int32_t rows_pad) {
    static const int env_cap = [] { const char *e = getenv("SAMD_GEMM_SPLIT_CAP"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
    static const int env_fixed = [] { const char *e = getenv("SAMD_GEMM_SPLITS"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
    const int cols = N / GEMM_COLS, chunks = K / GEMM_KC;
    (void)rows_pad;
    if (env_fixed) return env_fixed > chunks ? (chunks < 1 ? 1 : chunks) : env_fixed;
    int s = 256 / (cols > 0 ? cols : 1);
    if (s > env_cap) s = env_cap;
    if (s > chunks) s = chunks;
    return s < 1 ? 1 : s;
}

int64_t samd_gemm_workspace(int32_t rows_pad, int32_t N, int32_t splits) { return (int64_t)splits * rows_pad * N * 4; }

// out (dtype, [rows_pad][N]) when splits == 1, else fp32 partials [splits][rows_pad][N] in d_partial.
// rows_pad in {16, 32, 48, 64}; A must hold rows_pad rows (pad rows are read, their products land in pad rows).
int samd_gemm_skinny_silu(const void *d_A, const void *d_W, int32_t rows_pad, int32_t N, int32_t K, void *d_out, int32_t dtype, void *stream) {
    if (!d_A || !d_W || !d_out || (rows_pad != 16 && rows_pad != 32 && rows_pad != 48 && rows_pad != 64) || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC ||
        K % GEMM_KC != 0 || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_skinny_silu: unsupported shape (rows 16/32/48/64, N %% 128 == 0, K %% 256 == 0) or null pointer"); return SAMD_E_INVALID;
    }
    const hipError_t e = gemm_dispatch<1>(dtype, rows_pad, dim3(N / GEMM_COLS, 1), (hipStream_t)stream, d_A, d_W, nullptr, d_out, K, N, K / GEMM_KC, 1);
    if (e != hipSuccess) { samd_set_error("samd_gemm_skinny_silu: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_pack_weights(const void *d_W, void *d_packed, int32_t N, int32_t K, void *stream) {
    if (!d_W || !d_packed || d_W == d_packed || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC || K % GEMM_KC != 0) {
        samd_set_error("samd_gemm_pack_weights: needs N %% 128 == 0, K %% 256 == 0 and distinct buffers"); return SAMD_E_INVALID;
    }
    const long long units = (long long)N * K / 8;
    hipLaunchKernelGGL(k_gemm_pack, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_W, (uint4 *)d_packed, N, K);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_pack_groups(const void *d_W, void *d_packed, int32_t N, int32_t K, void *stream) {
    if (!d_W || !d_packed || d_W == d_packed || N < 16 || N % 16 != 0 || K < GEMM_KC || K % GEMM_KC != 0) {
        samd_set_error("samd_gemm_pack_groups: needs N %% 16 == 0, K %% 256 == 0 and distinct buffers"); return SAMD_E_INVALID;
    }
    const long long units = (long long)N * K / 8;
    hipLaunchKernelGGL(k_gemm_pack_groups, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_W, (uint4 *)d_packed, N, K);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_pairs_silu_norm(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_Wg, int32_t rows_pad, int32_t inter, int32_t K,
                              void *d_out, int32_t dtype, void *stream) {
    if (!d_x || !d_ssq || !d_norm_weight || !d_Wg || !d_out || (rows_pad != 16 && rows_pad != 8) || inter < 16 || inter % 16 != 0 || K < GEMM_KC || K % GEMM_KC != 0 ||
        K / 16 > 32 * NORM_NS || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_pairs_silu_norm: unsupported shape (8 or 16 rows, inter %% 16 == 0, K %% 256 == 0, K <= 8192, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    const int n_cu = samd_cu_count();
    const int n_pairs = inter / 16;
    int grid = n_pairs < n_cu ? n_pairs : n_cu;
    while ((n_pairs + grid - 1) / grid > 4) grid += n_cu;
    const NormArgs na{d_ssq, d_norm_weight, K / 16, 1.f / (float)K, eps};
    hipStream_t st = (hipStream_t)stream;
    // rows_pad 8: a draft of <= 8 nodes -- only rows 0..7 of x are fetched (the tile stays 16 rows; outputs of rows 8..15 are zero)
    const hipError_t e = rows_pad == 8
        ? (dtype == SAMD_F16 ? pairs_silu_launch<GF16, 1, 3, true, 8>(st, grid, d_x, d_Wg, d_out, K, inter, n_pairs, na)
                             : pairs_silu_launch<GBF16, 1, 3, true, 8>(st, grid, d_x, d_Wg, d_out, K, inter, n_pairs, na))
        : (dtype == SAMD_F16 ? pairs_silu_launch<GF16, 1, 3, true>(st, grid, d_x, d_Wg, d_out, K, inter, n_pairs, na)
                             : pairs_silu_launch<GBF16, 1, 3, true>(st, grid, d_x, d_Wg, d_out, K, inter, n_pairs, na));
    if (e != hipSuccess) { samd_set_error("samd_gemm_pairs_silu_norm: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_cs_residual(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t N, int32_t K, void *d_x, float *d_ssq, int32_t dtype, void *stream) {
    if (!d_A || !d_Wg || !d_x || !d_ssq || (rows_pad != 16 && rows_pad != 8) || N < 16 || N % 16 != 0 || K < GEMM_KC || K % GEMM_KC != 0 || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_cs_residual: unsupported shape (8 or 16 rows, N %% 16 == 0, K %% 256 == 0, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    constexpr int lds = GEMM_WAVES * 2 * 16 * GEMM_KC * 2;                       // 128 KiB: two A chunks per wave
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define GO(TT, ROWS, ET) do { static unsigned long long done = 0ull; \
        e = samd_reserve_lds((const void *)k_gemm_cs_residual<TT, ROWS>, lds, &done); \
        if (e == hipSuccess) hipLaunchKernelGGL((k_gemm_cs_residual<TT, ROWS>), dim3(N / 16), dim3(64 * GEMM_WAVES), lds, st, (const ET *)d_A, (const ET *)d_Wg, (ET *)d_x, d_ssq, K, N, K / GEMM_KC, (const int *)nullptr, (int *)nullptr, 0); } while (0)
    if (dtype == SAMD_F16) { if (rows_pad == 8) GO(GF16, 8, _Float16); else GO(GF16, 16, _Float16); }
    else { if (rows_pad == 8) GO(GBF16, 8, __bf16); else GO(GBF16, 16, __bf16); }
#undef GO
    if (e != hipSuccess) { samd_set_error("samd_gemm_cs_residual: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

/* seam experiment hook (scripts/seam_probe.py): samd_gemm_cs_residual for o_proj (K = 4096, <= 8 rows) launched on a SECOND stream beside the
 * attention launches; it requests its weights at entry and polls d_counter (samd_tree_attention_signal's arrivals) before it touches A.
 * d_epoch int32[N / 16], zero-initialised, owned by the caller; arrivals = n_q_pad * n_heads of the producing merge launch. */
int samd_gemm_cs_residual_early(const void *d_A, const void *d_Wg, int32_t N, int32_t K, void *d_x, float *d_ssq, int32_t dtype, const int32_t *d_counter,
                                int32_t *d_epoch, int32_t arrivals, void *stream) {
    if (!d_A || !d_Wg || !d_x || !d_ssq || !d_counter || !d_epoch || arrivals < 1 || N < 16 || N % 16 != 0 || K != 2 * GEMM_WAVES * GEMM_KC ||
        (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_cs_residual_early: unsupported shape (K must be 4096, N %% 16 == 0, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    constexpr int lds = GEMM_WAVES * 2 * 16 * GEMM_KC * 2;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (dtype == SAMD_F16) {
        static unsigned long long done = 0ull;
        e = samd_reserve_lds((const void *)k_gemm_cs_residual<GF16, 8, true>, lds, &done);
        if (e == hipSuccess) hipLaunchKernelGGL((k_gemm_cs_residual<GF16, 8, true>), dim3(N / 16), dim3(64 * GEMM_WAVES), lds, st, (const _Float16 *)d_A, (const _Float16 *)d_Wg,
                                                (_Float16 *)d_x, d_ssq, K, N, K / GEMM_KC, d_counter, d_epoch, arrivals);
    } else {
        static unsigned long long done = 0ull;
        e = samd_reserve_lds((const void *)k_gemm_cs_residual<GBF16, 8, true>, lds, &done);
        if (e == hipSuccess) hipLaunchKernelGGL((k_gemm_cs_residual<GBF16, 8, true>), dim3(N / 16), dim3(64 * GEMM_WAVES), lds, st, (const __bf16 *)d_A, (const __bf16 *)d_Wg,
                                                (__bf16 *)d_x, d_ssq, K, N, K / GEMM_KC, d_counter, d_epoch, arrivals);
    }
    if (e != hipSuccess) { samd_set_error("samd_gemm_cs_residual_early: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_pairs_silu(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t inter, int32_t K, void *d_out, int32_t dtype, void *stream) {
    if (!d_A || !d_Wg || !d_out || (rows_pad != 16 && rows_pad != 32 && rows_pad != 48 && rows_pad != 64) || inter < 16 || inter % 16 != 0 || K < GEMM_KC ||
        K % GEMM_KC != 0 || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_pairs_silu: unsupported shape (rows 16/32/48/64, inter %% 16 == 0, K %% 256 == 0, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    // one workgroup per CU (256 on MI355X) with an even share of the pairs; more workgroups only when a share would exceed 4 pairs (8 waves)
    const int n_cu = samd_cu_count();
    const int n_pairs = inter / 16;
    int grid = n_pairs < n_cu ? n_pairs : n_cu;
    while ((n_pairs + grid - 1) / grid > 4) grid += n_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    static const int depth_env = [] { const char *e = getenv("SAMD_PAIRS_DEPTH"); return e ? atoi(e) : 0; }();
#ifdef SAMD_GEMM_ABLATE
    static const int abl_set = [] { const char *e = getenv("SAMD_GEMM_ABL"); const int v = e ? atoi(e) : 0; return hipMemcpyToSymbol(HIP_SYMBOL(samd_abl_flag), &v, 4) == hipSuccess ? 1 : -1; }();
    (void)abl_set;
#endif
#define ARGS st, grid, d_A, d_Wg, d_out, K, inter, n_pairs
#define GO(TT) (rows_pad == 16 ? (depth_env == 2 ? pairs_silu_launch<TT, 1, 2>(ARGS) : depth_env == 4 ? pairs_silu_launch<TT, 1, 4>(ARGS) : pairs_silu_launch<TT, 1, 3>(ARGS)) \
                : rows_pad == 32 ? (depth_env == 2 ? pairs_silu_launch<TT, 2, 2>(ARGS) : pairs_silu_launch<TT, 2, 3>(ARGS)) \
                : rows_pad == 48 ? pairs_silu_launch<TT, 3, 2>(ARGS) : pairs_silu_launch<TT, 4, 2>(ARGS))
    e = dtype == SAMD_F16 ? GO(GF16) : GO(GBF16);
#undef GO
#undef ARGS
    if (e != hipSuccess) { samd_set_error("samd_gemm_pairs_silu: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_pack_qkv64(const void *d_W, void *d_packed, int32_t n_heads_total, int32_t K, void *stream) {
    if (!d_W || !d_packed || d_W == d_packed || n_heads_total < 1 || K < GEMM_KC || K % GEMM_KC != 0) {
        samd_set_error("samd_gemm_pack_qkv64: needs K %% 256 == 0, whole 128-column heads and distinct buffers"); return SAMD_E_INVALID;
    }
    const long long units = (long long)n_heads_total * 128 * K / 8;
    if (qkv_tile_groups(n_heads_total) == 3)
        hipLaunchKernelGGL(k_gemm_pack_qkv<3>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_W, (uint4 *)d_packed, n_heads_total * 128, K);
    else
        hipLaunchKernelGGL(k_gemm_pack_qkv<4>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_W, (uint4 *)d_packed, n_heads_total * 128, K);
    LAUNCHCHK();
    return SAMD_OK;
}

static int gemm_qkv_rope_impl(const void *d_A, const void *d_W64, int32_t rows_pad, int32_t K, const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n,
                              void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                              int32_t dtype, void *stream, int v_t) {
    if (!d_A || !d_W64 || !d_cs || !d_cache_length || !d_n || !d_q_out || !d_k_cache || !d_v_cache || head_dim != 128 || n_heads < 1 || n_kv_heads < 1 ||
        (rows_pad != 16 && rows_pad != 32 && rows_pad != 48 && rows_pad != 64) || K < GEMM_KC || K % GEMM_KC != 0 || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_qkv_rope: unsupported shape (rows 16/32/48/64, head_dim 128, K %% 256 == 0, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    static const int depth_env = [] { const char *e = getenv("SAMD_QKV_DEPTH"); return e ? atoi(e) : 0; }();
    const int groups = qkv_tile_groups(n_heads + 2 * n_kv_heads);
    const int tiles = (n_heads + 2 * n_kv_heads) * 128 / (16 * groups);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define GO(TT, RT, D, CG) e = qkv_rope_launch<TT, RT, D, CG>(st, d_A, d_W64, K, tiles, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_v_cache, n_heads, n_kv_heads, (long long)max_len, v_t)
#define ROWS(TT, CG) do { if (rows_pad == 16) { if (depth_env == 2) GO(TT, 1, 2, CG); else if (depth_env == 3) GO(TT, 1, 3, CG); else GO(TT, 1, 4, CG); } \
                          else if (rows_pad == 32) { if (depth_env == 2) GO(TT, 2, 2, CG); else GO(TT, 2, 4, CG); } \
                          else if (rows_pad == 48) GO(TT, 3, 3, CG); else GO(TT, 4, 3, CG); } while (0)
    if (groups == 3) { if (dtype == SAMD_F16) ROWS(GF16, 3); else ROWS(GBF16, 3); }
    else { if (dtype == SAMD_F16) ROWS(GF16, 4); else ROWS(GBF16, 4); }
#undef ROWS
#undef GO
    if (e != hipSuccess) { samd_set_error("samd_gemm_qkv_rope: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_qkv_rope(const void *d_A, const void *d_W64, int32_t rows_pad, int32_t K, const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n,
                       void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                       int32_t dtype, void *stream) {
    return gemm_qkv_rope_impl(d_A, d_W64, rows_pad, K, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_v_cache, n_heads, n_kv_heads, head_dim, max_len, dtype, stream, 0);
}

/* round 6: the same with the V rows written into a TRANSPOSED cache, d_vt_cache [H_kv][128][max_len] (what samd_tree_attention_vt reads) */
int samd_gemm_qkv_rope_vt(const void *d_A, const void *d_W64, int32_t rows_pad, int32_t K, const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n,
                          void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                          int32_t dtype, void *stream) {
    return gemm_qkv_rope_impl(d_A, d_W64, rows_pad, K, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_vt_cache, n_heads, n_kv_heads, head_dim, max_len, dtype, stream, 1);
}

static int gemm_qkv_rope_norm_impl(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_W64, int32_t rows_pad, int32_t K,
                                   const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_v_cache,
                                   int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype, void *stream, int v_t) {
    if (!d_x || !d_ssq || !d_norm_weight || !d_W64 || !d_cs || !d_cache_length || !d_n || !d_q_out || !d_k_cache || !d_v_cache || head_dim != 128 || n_heads < 1 ||
        n_kv_heads < 1 || (rows_pad != 16 && rows_pad != 8) || K < GEMM_KC || K % GEMM_KC != 0 || K / 16 > 32 * NORM_NS || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_qkv_rope_norm: unsupported shape (8 or 16 rows, head_dim 128, K %% 256 == 0, K <= 8192, f16/bf16) or null pointer"); return SAMD_E_INVALID;
    }
    const int groups = qkv_tile_groups(n_heads + 2 * n_kv_heads);
    const int tiles = (n_heads + 2 * n_kv_heads) * 128 / (16 * groups);
    const NormArgs na{d_ssq, d_norm_weight, K / 16, 1.f / (float)K, eps};
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define GO(TT, CG, AR) e = qkv_rope_launch<TT, 1, 4, CG, true, AR>(st, d_x, d_W64, K, tiles, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_v_cache, n_heads, n_kv_heads, (long long)max_len, v_t, na)
    // rows_pad 8: a draft of <= 8 nodes -- only rows 0..7 of x are fetched (the tile stays 16 rows)
    if (rows_pad == 8) {
        if (groups == 3) { if (dtype == SAMD_F16) GO(GF16, 3, 8); else GO(GBF16, 3, 8); }
        else { if (dtype == SAMD_F16) GO(GF16, 4, 8); else GO(GBF16, 4, 8); }
    } else {
        if (groups == 3) { if (dtype == SAMD_F16) GO(GF16, 3, 16); else GO(GBF16, 3, 16); }
        else { if (dtype == SAMD_F16) GO(GF16, 4, 16); else GO(GBF16, 4, 16); }
    }
#undef GO
    if (e != hipSuccess) { samd_set_error("samd_gemm_qkv_rope_norm: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_gemm_qkv_rope_norm(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_W64, int32_t rows_pad, int32_t K,
                            const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_v_cache,
                            int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype, void *stream) {
    return gemm_qkv_rope_norm_impl(d_x, d_ssq, d_norm_weight, eps, d_W64, rows_pad, K, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_v_cache, n_heads, n_kv_heads, head_dim,
                                   max_len, dtype, stream, 0);
}

int samd_gemm_qkv_rope_norm_vt(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_W64, int32_t rows_pad, int32_t K,
                               const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_vt_cache,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype, void *stream) {
    return gemm_qkv_rope_norm_impl(d_x, d_ssq, d_norm_weight, eps, d_W64, rows_pad, K, d_cs, d_cache_length, d_n, d_q_out, d_k_cache, d_vt_cache, n_heads, n_kv_heads, head_dim,
                                   max_len, dtype, stream, 1);
}

int samd_gemm_skinny(const void *d_A, const void *d_W, int32_t rows_pad, int32_t N, int32_t K, int32_t splits, float *d_partial,
                     void *d_out, int32_t dtype, void *stream) {
    if (!d_A || !d_W || (rows_pad != 16 && rows_pad != 32 && rows_pad != 48 && rows_pad != 64) || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC ||
        K % GEMM_KC != 0 || splits < 1 || splits > K / GEMM_KC || (splits == 1 ? !d_out : !d_partial) || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_skinny: unsupported shape (rows 16/32/48/64, N %% 128 == 0, K %% 256 == 0) or null pointer"); return SAMD_E_INVALID;
    }
    const hipError_t e = gemm_dispatch<0>(dtype, rows_pad, dim3(N / GEMM_COLS, splits), (hipStream_t)stream, d_A, d_W, d_partial, splits == 1 ? d_out : nullptr, K, N,
                                          K / GEMM_KC, splits);
    if (e != hipSuccess) { samd_set_error("samd_gemm_skinny: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

/* samd_gemm_skinny over a GROUP-MAJOR matrix (samd_gemm_pack_groups of the plain [N][K] weight: the layout samd_gemm_cs_residual reads), so that
 * o_proj / down_proj keep one packed copy for every row bucket.  Same arguments, same results bit for bit (the lanes multiply the same values in
 * the same order; only where a wave finds its 16 columns differs). */
int samd_gemm_skinny_groups(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t N, int32_t K, int32_t splits, float *d_partial,
                            void *d_out, int32_t dtype, void *stream) {
    if (!d_A || !d_Wg || (rows_pad != 16 && rows_pad != 32 && rows_pad != 48 && rows_pad != 64) || N < GEMM_COLS || N % GEMM_COLS != 0 || K < GEMM_KC ||
        K % GEMM_KC != 0 || splits < 1 || splits > K / GEMM_KC || (splits == 1 ? !d_out : !d_partial) || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_gemm_skinny_groups: unsupported shape (rows 16/32/48/64, N %% 128 == 0, K %% 256 == 0) or null pointer"); return SAMD_E_INVALID;
    }
    const hipError_t e = gemm_dispatch<0, true>(dtype, rows_pad, dim3(N / GEMM_COLS, splits), (hipStream_t)stream, d_A, d_Wg, d_partial, splits == 1 ? d_out : nullptr, K, N,
                                                K / GEMM_KC, splits);
    if (e != hipSuccess) { samd_set_error("samd_gemm_skinny_groups: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
