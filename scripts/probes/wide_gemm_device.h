// wide_gemm_device.h -- EXPERIMENT of round 5, not part of the library (profiles/r05_wide_gemm.md: on the layer's shapes it loses to hipBLASLt once
// its tiles have to be balanced over 256 CUs).  The compute-bound GEMM of the request start (prompt prefill, M = 65..2048 rows):
//     C[M][N] = A[M][K] x W[N][K]^T        (W = an nn.Linear weight as HF stores it; what the reference runs through HF's
//                                            q/k/v/o/gate/up/down projections on the prompt, SO/samd_model.py:102-106)
// hand-written for gfx950 instead of hipBLASLt's 0.5-1.0 PFLOP/s on these shapes (profiles/r05_prefill.md: 256 CUs against 96-516 tiles).
//
// Tile: 256 weight rows x 256 activation rows per workgroup of 8 waves (2 x 4), each wave 128 x 64 outputs = 8 x 4 fragments of
// v_mfma_f32_16x16x32 (weights = the MFMA's row operand, so a lane ends up with 4 consecutive n of one m: 8-byte stores into C[m][n]).
// k-tiles of 64 (whole 128-byte lines of every row) through a ring of eight 16 KiB LDS UNITS = two k-tiles x {W-lo, A-lo, A-hi, W-hi}, filled
// by LDS-DMA (global_load_lds_dwordx4, scalar base + lane offset) with the bank swizzle on the SOURCE address: unit row r, 16-byte slot s is
// stored at slot s ^ ((r >> 1) & 7) -- ds_read_b128's four 16-lane groups then touch 16 different slots of the 256-byte bank row.
//
// The k loop runs in PHASES of 16 MFMAs (one quadrant of the wave's outputs over the k-tile), four per k-tile, each a LOAD half (barrier;
// ds_reads of the next phase's new fragments; ONE unit's LDS-DMA for the k-tile two ahead into the slot whose readers are done; counted
// lgkmcnt / vmcnt waits) and a COMPUTE half (barrier; 16 MFMAs).  The second wave of every SIMD (w >= 4) runs one barrier behind the
// first, so one wave issues its reads while the other keeps the MFMA pipe busy; a barrier then certifies for the other group only what a
// wave did half a phase earlier, which is why both waits sit at the end of the load half.  Five units (80 KiB) are in flight per CU.
//   p1: W-lo x A-lo   reads A-hi(t)           stages W-hi(t+1)
//   p2: W-lo x A-hi   reads W-hi(t)           stages W-lo(t+2)
//   p3: W-hi x A-hi   --                      stages A-lo(t+2)
//   p4: W-hi x A-lo   reads W-lo, A-lo(t+1)   stages A-hi(t+2)
// Measured (scripts/probes/wide_gemm_probe.hip, profiles/r05_wide_gemm.md): 1.08-1.22 PFLOP/s on full rounds of tiles (one CU alone: 1.07-1.25 us
// per k-tile = 0.7-0.8 of the MFMA peak; with 256 CUs 1.7-2.0 us, bound by what the L2s deliver: 38 GB/s per CU).
//
// WORK is dealt out stream-K: the (tile, k-tile) iteration space, tiles in XCD-contiguous order with the activation tile fastest, is cut
// into one contiguous range per workgroup of a persistent grid (one per CU).  A range = [tail of a tile] + whole tiles + [head of a tile];
// a partly covered tile's fp32 accumulators go to a workspace slot, a per-tile counter collects the contributors (agent-scope release /
// acquire, MI355X_MICROARCH.md "inter-workgroup visibility"), and when all c of them have arrived contributor r adds up, in contributor
// order, and stores the r-th share of the tile -- after ALL its own compute, so nobody waits while somebody else needs its CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace widegemm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct F16 { typedef _Float16 elem; typedef half8 vec8; typedef _Float16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
struct BF16 { typedef __bf16 elem; typedef bf16x8 vec8; typedef __bf16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };

#ifndef WIDE_SC1
#define WIDE_SC1 1                // partial tiles: 1 = agent-scope (sc1) write-through stores and sc1 loads, no fences; 0 = plain stores + release / acquire fences
#endif
#ifndef WIDE_KSTEP
#define WIDE_KSTEP 128            // bytes per row per k-tile (-DWIDE_KSTEP=0: timing experiment, every k-tile re-reads the first)
#endif
constexpr int WT = 256;           // weight rows (output columns) per tile
constexpr int AT = 256;           // activation rows per tile
constexpr int BK = 64;            // k per k-tile
constexpr int RING_BYTES = 131072;
constexpr int U_WLO = 0;          // weight rows wn * 128 + 0..63    (fragments i = 0..3)
constexpr int U_ALO = 16384;      // activation rows wm * 64 + 0..31 (fragments j = 0, 1)
constexpr int U_AHI = 32768;      // activation rows wm * 64 + 32..63 (fragments j = 2, 3)
constexpr int U_WHI = 49152;      // weight rows wn * 128 + 64..127  (fragments i = 4..7)
constexpr int SLOT_FLOATS = 256 * 256;                 // one partial tile
constexpr int EPI_PLAIN = 0;      // C[m][n]
constexpr int EPI_SILU = 1;       // W = [gate rows; up rows] (N = 2 x inter): C[m][o] = silu(gate) * up, [M][inter] -- LlamaMLP's act_fn(gate_proj(x)) * up_proj(x)

struct Ctx {
    floatx4 (&acc)[8][4];
    u32x4 wlo[8], whi[8], ahi[4], alo[2][4];     // [2 * fragment + k half]
    // Per-lane addressing state is THREE registers; everything else is derived from them per use with wave-uniform terms (a handful of VALU
    // operations per k-tile): more live VGPRs spill inside the k loop, and a scratch reload there counts on vmcnt.
    uint32_t aW0;                                // LDS read address of W fragment 0, k half 0, k-tile buffer 0 (k half 1: ^ 64; A: + a_delta; buffer 1: + 65536)
    uint32_t vW0, vA0;                           // byte offsets of this lane's first staging source row of a W-lo / A-lo unit
    uint32_t a_delta;                            // uniform: A fragment 0 relative to W fragment 0 in a unit
    uint32_t w_step, a_row_bytes, a_last;        // uniform: the lane's second W row (wave row 1) in bytes; one row of A; byte offset of A's last row
    const char *gW, *gA;                         // uniform: the tile's weight rows / the activation matrix, at the k-tile being staged
    size_t whi_off;                              // W-hi rows relative to W-lo rows (64 rows; EPI_SILU: the up matrix)
    uint32_t lds_base, wave_dst;
    __device__ __forceinline__ Ctx(floatx4 (&a)[8][4]) : acc(a) {}
};

// One unit = two LDS-DMA wave-instructions per wave (8 unit rows x 128 B each), hand-issued in the scalar-base form (the builtin wants a
// 64-bit address per lane: 12 VGPRs this kernel does not have).  M0 = the wave's LDS destination; the hardware adds lane * 16.
template <int B, int UOFF> __device__ __forceinline__ void stage_unit(Ctx &c, const char *g, uint32_t v0, uint32_t v1) {
    const uint32_t dst0 = c.lds_base + B * 65536 + UOFF + c.wave_dst, dst1 = dst0 + 8192;
    // s_mov only: an s_add on M0 would also write SCC, which the compiler may be holding a loop condition in (it does not look inside the asm)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 : : "v"(v0), "v"(v1), "s"(g), "s"(dst0), "s"(dst1) : "memory");   // (M0 is not on the clobber list: the compiler reserves it and uses it for nothing else in this kernel)
}
// (the empty asm makes the derived value opaque: the compiler would otherwise hoist it out of the k loop as an invariant and spill it there)
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
template <int B, int UOFF> __device__ __forceinline__ void stage_w(Ctx &c, const char *g) { const uint32_t v = opaque(c.vW0); stage_unit<B, UOFF>(c, g, v, v + c.w_step); }
// activation rows: the lane's second row sits 128 rows further, A-hi rows 32 further; rows past M are clamped to the last row (they may read
// any valid row: their outputs are never stored)
template <int B, int UOFF> __device__ __forceinline__ void stage_a(Ctx &c, const char *g) {
    const uint32_t base = opaque(c.vA0), cap = c.a_last + (base & 127);
    uint32_t v0 = base + (UOFF == U_AHI ? 32 : 0) * c.a_row_bytes, v1 = base + (UOFF == U_AHI ? 160 : 128) * c.a_row_bytes;
    v0 = v0 < cap ? v0 : cap; v1 = v1 < cap ? v1 : cap;
    stage_unit<B, UOFF>(c, g, v0, v1);
}
#define WIDE_RD(o, a, off) "ds_read_b128 %" #o ", %" #a " offset:" #off "\n\t"
template <int B> __device__ __forceinline__ void read_lo(Ctx &c, u32x4 (&alo)[4]) {        // 8 W-lo + 4 A-lo
    const uint32_t w0 = opaque(c.aW0) + B * 65536, w1 = w0 ^ 64, a0 = w0 + c.a_delta, a1 = a0 ^ 64;
    asm volatile(WIDE_RD(0, 12, 0) WIDE_RD(1, 13, 0) WIDE_RD(8, 14, 16384) WIDE_RD(9, 15, 16384) WIDE_RD(2, 12, 2048) WIDE_RD(3, 13, 2048)
                 WIDE_RD(10, 14, 18432) WIDE_RD(11, 15, 18432) WIDE_RD(4, 12, 4096) WIDE_RD(5, 13, 4096) WIDE_RD(6, 12, 6144) WIDE_RD(7, 13, 6144)
                 : "=&v"(c.wlo[0]), "=&v"(c.wlo[1]), "=&v"(c.wlo[2]), "=&v"(c.wlo[3]), "=&v"(c.wlo[4]), "=&v"(c.wlo[5]), "=&v"(c.wlo[6]), "=&v"(c.wlo[7]),
                   "=&v"(alo[0]), "=&v"(alo[1]), "=&v"(alo[2]), "=&v"(alo[3])
                 : "v"(w0), "v"(w1), "v"(a0), "v"(a1));
}
template <int B> __device__ __forceinline__ void read_ahi(Ctx &c) {
    const uint32_t a0 = opaque(c.aW0) + B * 65536 + c.a_delta, a1 = a0 ^ 64;
    asm volatile(WIDE_RD(0, 4, 32768) WIDE_RD(1, 5, 32768) WIDE_RD(2, 4, 34816) WIDE_RD(3, 5, 34816)
                 : "=&v"(c.ahi[0]), "=&v"(c.ahi[1]), "=&v"(c.ahi[2]), "=&v"(c.ahi[3]) : "v"(a0), "v"(a1));
}
template <int B> __device__ __forceinline__ void read_whi(Ctx &c) {
    const uint32_t w0 = opaque(c.aW0) + B * 65536, w1 = w0 ^ 64;
    asm volatile(WIDE_RD(0, 8, 49152) WIDE_RD(1, 9, 49152) WIDE_RD(2, 8, 51200) WIDE_RD(3, 9, 51200) WIDE_RD(4, 8, 53248) WIDE_RD(5, 9, 53248)
                 WIDE_RD(6, 8, 55296) WIDE_RD(7, 9, 55296)
                 : "=&v"(c.whi[0]), "=&v"(c.whi[1]), "=&v"(c.whi[2]), "=&v"(c.whi[3]), "=&v"(c.whi[4]), "=&v"(c.whi[5]), "=&v"(c.whi[6]), "=&v"(c.whi[7])
                 : "v"(w0), "v"(w1));
}
#undef WIDE_RD
// "every ds_read but the youngest N has returned", tied to the registers it covers so that the MFMAs reading them stay behind it
template <int N> __device__ __forceinline__ void lds_wait8(u32x4 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
}
template <int N> __device__ __forceinline__ void lds_wait4(u32x4 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}
template <typename TT, int I0, int J0> __device__ __forceinline__ void mfma_quadrant(Ctx &c, const u32x4 (&wf)[8], const u32x4 (&af)[4]) {
    typedef typename TT::vec8 V8;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                c.acc[I0 + i][J0 + j] = TT::mfma(__builtin_bit_cast(V8, wf[2 * i + ks]), __builtin_bit_cast(V8, af[2 * j + ks]), c.acc[I0 + i][J0 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
}
// the four younger units stay in flight; once a stage has been skipped (the k range ends) the count no longer holds: drain
__device__ __forceinline__ void vm_wait(bool steady) {
    if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void half_open() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename TT, int B> __device__ __forceinline__ void ktile(Ctx &c, int t, int nk) {
    const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
    half_open();                                                                     // p1
    read_ahi<B>(c);
    if (has1) stage_w<B ^ 1, U_WHI>(c, c.gW + c.whi_off - WIDE_KSTEP);     // k-tile t + 1 (the running pointers stand at t + 2)
    lds_wait8<4>(c.wlo); lds_wait4<4>(c.alo[B]);
    vm_wait(has2);
    half_open();
    mfma_quadrant<TT, 0, 0>(c, c.wlo, c.alo[B]);
    half_open();                                                                     // p2
    read_whi<B>(c);
    if (has2) stage_w<B, U_WLO>(c, c.gW);
    lds_wait4<8>(c.ahi);
    vm_wait(has2);
    half_open();
    mfma_quadrant<TT, 0, 2>(c, c.wlo, c.ahi);
    half_open();                                                                     // p3
    if (has2) stage_a<B, U_ALO>(c, c.gA);
    lds_wait8<0>(c.whi);
    vm_wait(has2);
    half_open();
    mfma_quadrant<TT, 4, 2>(c, c.whi, c.ahi);
    half_open();                                                                     // p4
    if (has1) read_lo<B ^ 1>(c, c.alo[B ^ 1]);
    if (has2) stage_a<B, U_AHI>(c, c.gA);
    if (has1) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");                   // nothing of this phase is outstanding; the other group's stager relies on it
    vm_wait(has2);
    half_open();
    mfma_quadrant<TT, 4, 0>(c, c.whi, c.alo[B]);
    c.gW += WIDE_KSTEP; c.gA += WIDE_KSTEP;
}

// acc += Wtile[256 rows][k-tiles kt0..kt1) x Atile[256 rows][same k)^T.  Wlo = the tile's first weight row (of the W-lo half of wave row 0);
// w_stride = rows between the two wave rows' W-lo halves (128; EPI_SILU 64); whi_rows = rows from a W-lo row to its W-hi row (64; EPI_SILU: inter).
template <typename TT>
__device__ __forceinline__ void fragment(const typename TT::elem *__restrict__ A, const typename TT::elem *__restrict__ Wlo, int M, int K, int m0,
                                         int w_stride, size_t whi_rows, int kt0, int kt1, floatx4 (&acc)[8][4], uint32_t lds_base) {
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, wn = w & 1, wm = w >> 1;
    const int nk = kt1 - kt0;
    Ctx c(acc);
    c.lds_base = lds_base;
    c.wave_dst = (uint32_t)w * 1024;
    // staging: a wave-instruction fills 8 unit rows x 128 B; the lane's rows are unit rows r = 8 w + (l >> 3) and r + 64; LDS slot l & 7 holds
    // SOURCE slot (l & 7) ^ ((r >> 1) & 7) (the same for both rows).  W unit row r = wave row (r >> 6), row (r & 63); A: wave row (r >> 5), row (r & 31)
    {
        const int r = 8 * w + (l >> 3), s = (l & 7) ^ ((r >> 1) & 7);
        c.vW0 = (uint32_t)r * (uint32_t)K * 2 + 16 * s;
        int g0 = m0 + (r >> 5) * 64 + (r & 31);
        g0 = g0 < M ? g0 : M - 1;
        c.vA0 = (uint32_t)g0 * (uint32_t)K * 2 + 16 * s;
    }
    c.w_step = (uint32_t)w_stride * (uint32_t)K * 2;
    c.a_row_bytes = (uint32_t)K * 2;
    c.a_last = (uint32_t)(M - 1) * (uint32_t)K * 2;
    c.whi_off = whi_rows * (size_t)K * 2;
    c.gW = reinterpret_cast<const char *>(Wlo) + (size_t)kt0 * 128;
    c.gA = reinterpret_cast<const char *>(A) + (size_t)kt0 * 128;
    // fragment reads: lane (l & 15) = row within a 16-row fragment, l >> 4 = 16-byte k slot within the k half (k half 1: slot + 4, i.e. the
    // swizzled byte address ^ 64); fragment i sits 2048 B further
    c.aW0 = lds_base + (uint32_t)(wn * 64 + (l & 15)) * 128 + (uint32_t)(((l >> 4) ^ ((l & 15) >> 1)) * 16);
    c.a_delta = (uint32_t)((wm * 32 - wn * 64) * 128);
    // prologue: k-tile 0 whole, k-tile 1 but its W-hi unit (p1 of k-tile 0 stages that one)
    stage_w<0, U_WLO>(c, c.gW);
    stage_a<0, U_ALO>(c, c.gA);
    stage_a<0, U_AHI>(c, c.gA);
    stage_w<0, U_WHI>(c, c.gW + c.whi_off);
    c.gW += WIDE_KSTEP; c.gA += WIDE_KSTEP;
    if (nk >= 2) {
        stage_w<1, U_WLO>(c, c.gW);
        stage_a<1, U_ALO>(c, c.gA);
        stage_a<1, U_AHI>(c, c.gA);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    c.gW += WIDE_KSTEP; c.gA += WIDE_KSTEP;                                   // the running pointers stand at k-tile t + 2
    half_open();
    read_lo<0>(c, c.alo[0]);
    if (nk >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (w >= 4) half_open();                                                  // the second group runs one barrier behind ...
    for (int t = 0; t < nk; t += 2) {
        ktile<TT, 0>(c, t, nk);
        if (t + 1 < nk) ktile<TT, 1>(c, t + 1, nk);
    }
    if (w < 4) half_open();                                                   // ... and the first one waits for it here
    half_open();                                                              // the ring is free for the next fragment's prologue
}

__device__ __forceinline__ void zero_acc(floatx4 (&acc)[8][4]) {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
}

// the (i, j) fragment pair {acc[i][j], acc[i + 4][j]} -> C.  EPI_PLAIN: C[m][n0 + wn * 128 + (i | i + 4) * 16 + 4 g + r];
// EPI_SILU: out[m][o0 + wn * 64 + i * 16 + 4 g + r] = silu(gate) * up with HF's roundings (act_fn in the model dtype, then the product)
template <typename TT, int EPI>
__device__ __forceinline__ void store_pair(typename TT::elem *__restrict__ C, int ldc, int m, int col0, int i, int wn, int g, const floatx4 &lo, const floatx4 &hi) {
    typedef typename TT::elem E;
    typedef typename TT::vec4 V4;
    if constexpr (EPI == EPI_SILU) {
        V4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float gf = (float)(E)lo[r], uf = (float)(E)hi[r];
            const E sv = (E)(gf / (1.f + __expf(-gf)));
            v[r] = (E)((float)sv * uf);
        }
        *reinterpret_cast<V4 *>(C + (size_t)m * ldc + col0 + wn * 64 + i * 16 + 4 * g) = v;
    } else {
        V4 a = {(E)lo[0], (E)lo[1], (E)lo[2], (E)lo[3]}, b = {(E)hi[0], (E)hi[1], (E)hi[2], (E)hi[3]};
        E *p = C + (size_t)m * ldc + col0 + wn * 128 + i * 16 + 4 * g;
        *reinterpret_cast<V4 *>(p) = a;
        *reinterpret_cast<V4 *>(p + 64) = b;
    }
}

template <typename TT, int EPI>
__device__ __forceinline__ void store_tile(typename TT::elem *__restrict__ C, int M, int ldc, int col0, int m0, const floatx4 (&acc)[8][4]) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, wn = w & 1, wm = w >> 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = m0 + wm * 64 + j * 16 + (l & 15);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; i++) store_pair<TT, EPI>(C, ldc, m, col0, i, wn, l >> 4, acc[i][j], acc[i + 4][j]);
    }
}

// XCD-aware order: hardware deals workgroup b to XCD b % 8; give every XCD a contiguous run of logical positions
__device__ __forceinline__ int xcd_logical(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct Args {
    const void *A, *W;
    void *C;
    float *slots;             // [grid][2][SLOT_FLOATS] fp32 partial tiles
    int *counters;            // [tiles], zero between launches (the kernel leaves them zero)
    int *fault;               // set to 1 when a wait ran out (a lost contributor): the result is then invalid
    int M, N, K, tiles_m, tiles_n, KT, ldc;
    long long iters;          // tiles x KT
};

__device__ __forceinline__ long long range_begin(long long iters, int grid, int pos) { return iters * pos / grid; }
// the grid position whose range holds iteration `it`
__device__ __forceinline__ int range_owner(long long iters, int grid, long long it) {
    int p = (int)(it * grid / iters);
    while (p + 1 < grid && range_begin(iters, grid, p + 1) <= it) p++;
    while (p > 0 && range_begin(iters, grid, p) > it) p--;
    return p;
}

template <typename TT, int EPI>
__global__ __launch_bounds__(512, 1) void k_wide_gemm(Args a) {
    typedef typename TT::elem E;
    extern __shared__ __attribute__((aligned(1024))) char wide_lds[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)wide_lds;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int grid = gridDim.x, pos = xcd_logical(blockIdx.x, grid);
    const long long it_begin = range_begin(a.iters, grid, pos), it_end = range_begin(a.iters, grid, pos + 1);
    const E *A = reinterpret_cast<const E *>(a.A), *W = reinterpret_cast<const E *>(a.W);
    E *C = reinterpret_cast<E *>(a.C);
    const int w_stride = EPI == EPI_SILU ? 64 : 128, w_tile = EPI == EPI_SILU ? 128 : 256;
    const size_t whi_rows = EPI == EPI_SILU ? (size_t)a.N / 2 : 64;
    int part_tile[2] = {-1, -1};
    long long it = it_begin;
    while (it < it_end) {
        const int tile = (int)(it / a.KT), kt0 = (int)(it % a.KT);
        const long long left = it_end - it;
        const int kt1 = left < a.KT - kt0 ? kt0 + (int)left : a.KT;
        const int tn = tile / a.tiles_m, tm = tile % a.tiles_m;
        floatx4 acc[8][4];
        zero_acc(acc);
        fragment<TT>(A, W + (size_t)tn * w_tile * a.K, a.M, a.K, tm * AT, w_stride, whi_rows, kt0, kt1, acc, lds_base);
        if (kt0 == 0 && kt1 == a.KT) store_tile<TT, EPI>(C, a.M, a.ldc, tn * w_tile, tm * AT, acc);
        else {
            // my first fragment -> slot 0; a later partial one (it can only be my last) -> slot 1
            const int which = it == it_begin ? 0 : 1;
            float *slot = a.slots + ((size_t)pos * 2 + which) * SLOT_FLOATS + ((size_t)w * 32 * 64 + l) * 4;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float *p = slot + (size_t)(i * 4 + j) * 256;
                    if (WIDE_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(acc[i][j]) : "memory");
                    else *reinterpret_cast<floatx4 *>(p) = acc[i][j];
                }
            part_tile[which] = tile;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (!WIDE_SC1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_fetch_add(&a.counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        it += kt1 - kt0;
    }
    // my shares of the tiles I hold a part of
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
        const int tile = part_tile[which];
        if (tile < 0) continue;
        const long long t_begin = (long long)tile * a.KT;
        const int first = range_owner(a.iters, grid, t_begin), last = range_owner(a.iters, grid, t_begin + a.KT - 1);
        const int c = last - first + 1, rank = pos - first;
        if (tid == 0) {
            long long spins = 0;
            while (__hip_atomic_load(&a.counters[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1ll << 24)) { *a.fault = 1; break; }
            }
            if (!WIDE_SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const int tn = tile / a.tiles_m, tm = tile % a.tiles_m, wn = w & 1, wm = w >> 1;
        const int shares = c < 16 ? c : 16;
        if (rank < shares) {
            for (int u = rank; u < 16; u += shares) {                       // unit u = fragments (i, j) and (i + 4, j), i = u >> 2, j = u & 3
                const int i = u >> 2, j = u & 3;
                floatx4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
                for (int q = 0; q < c; q++) {
                    const int p = first + q;
                    const int sw = range_begin(a.iters, grid, p) < t_begin ? 1 : 0;   // the tile is p's last fragment unless p's range starts inside it
                    const float *s = a.slots + ((size_t)p * 2 + sw) * SLOT_FLOATS + ((size_t)w * 32 * 64 + l) * 4;
                    floatx4 x, y;
                    if (WIDE_SC1) {
                        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(x), "=&v"(y) : "v"(s + (size_t)(i * 4 + j) * 256), "v"(s + (size_t)((i + 4) * 4 + j) * 256) : "memory");
                    } else {
                        x = *reinterpret_cast<const floatx4 *>(s + (size_t)(i * 4 + j) * 256);
                        y = *reinterpret_cast<const floatx4 *>(s + (size_t)((i + 4) * 4 + j) * 256);
                    }
                    lo += x; hi += y;
                }
                const int m = tm * AT + wm * 64 + j * 16 + (l & 15);
                if (m < a.M) store_pair<TT, EPI>(C, a.ldc, m, tn * w_tile, i, wn, l >> 4, lo, hi);
            }
        }
        // second count: the last reader puts the counter back to zero for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&a.counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 2 * c - 1) __hip_atomic_store(&a.counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace widegemm
