// attn_kernels.hip -- the attention block of one decoder layer of the verify forward as ONE launch on gfx950:
//   RoPE on q and k, the K / V row write of SamdStaticCache.update (SO/cache.py:103-115) at [Lw, Lw + n), tree-mask attention of
//   the n draft rows over the cached keys + the n new ones (SO/model_patch/llama.py:82-96 + the SDPA call it feeds), and the
//   merge of the per-tile partial softmaxes.
// The three-launch form (k_rope_kv -> k_tree_attention over 16 KV splits -> k_attn_combine, verify_kernels.hip / lm_kernels.hip)
// spends most of its 17-19 us per layer on launch boundaries and on memory round trips that depend on each other; folding the
// merge into the split kernel with a cross-workgroup arrival counter measured SLOWER than the extra launch (profiles/
// r02_attention_variants.md).  What works is to never need a cross-workgroup merge:
//
//   * ONE workgroup per (query head, block of 16 query rows), 8 wavefronts; wave w owns the 64-key tiles w, w + 8, ... of the
//     cached keys, wave 7 additionally the tile of the n new keys.  Every wave runs QK^T -> online softmax -> PV on its tiles
//     with no workgroup barrier in the loop; the 8 partial (m, l, O) meet in LDS once and are merged by all 512 threads.
//   * V is cached TRANSPOSED ([H_kv][D][max_len], "V^T"): the PV product's B operand (16 d-columns x 32 keys, 8 consecutive keys
//     per lane) is then a plain 16-byte global load, so V needs no LDS staging, no transposition and no barrier.  K stays
//     [H_kv][max_len][D] (already the QK^T B-operand layout).
//   * everything a wave needs first -- its Q rows (from the q|k|v projection's output or fp32 split-K partials, + RoPE from
//     per-row cos/sin prepared once per forward by k_rope_rows), its first K / V^T tile, the mask rows, L and n -- is requested at
//     kernel entry in one batch; the new keys belong to a wave known from the block index, so their source rows are in that batch.
//   * visible prefix: keys < Lvis are visible to every row; key Lvis + j is visible to row i iff bit j of mask[i].  The base
//     model's verify has Lvis = Lw = L.  A draft head's tree level keeps earlier levels' rows in its cache: Lvis = accepted
//     length, Lw beyond it (the bits then address the earlier levels' rows and the new ones alike).
// Per workgroup at most ceil(L / 512) tile rounds; a 1000-key context is two.  (Contexts of many thousands of keys would want a
// second split level again; Llama-3's 8192 positions cost 16 rounds here.)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "samd_common.h"

#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct AF16 { typedef _Float16 elem; typedef half8 vec8;
    static __device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
struct ABF16 { typedef __bf16 elem; typedef bf16x8 vec8;
    static __device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };

#define AB_D 128
#define AB_TILE 64
#define AB_WAVES 8
#define AB_PSTRIDE (AB_TILE + 8)       // halfs per P row
#define AB_OSTRIDE (AB_D + 4)          // floats per merged-O row
#define AB_QSTRIDE (AB_D + 8)          // halfs per staged Q / new-K / new-V row: 272 B, 16 rows of one ds_read_b128 hit different banks
#define AB_LDS_QS (AB_WAVES * 16 * AB_PSTRIDE * 2)
#define AB_LDS_KN (AB_LDS_QS + 16 * AB_QSTRIDE * 2)
#define AB_LDS_VN (AB_LDS_KN + AB_TILE * AB_QSTRIDE * 2)
#define AB_LDS_OM (AB_LDS_VN + AB_TILE * AB_QSTRIDE * 2)
#define AB_LDS_MM (AB_LDS_OM + AB_WAVES * 16 * AB_OSTRIDE * 4)
#define AB_LDS_BYTES (AB_LDS_MM + 2 * AB_WAVES * 16 * 4)

// cos / sin of every row's position (base + relative position), fp32 [rows][128]: cos in [0, 64), sin in [64, 128)
__global__ __launch_bounds__(256) void k_rope_rows(const int *__restrict__ rel_pos, const int *__restrict__ d_base, const float *__restrict__ cos_t,
                                                   const float *__restrict__ sin_t, float *__restrict__ cs, int rows, int max_pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = i >> 6, j = i & 63;
    if (r >= rows) return;
    int pos = d_base[0] + rel_pos[r];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    cs[r * AB_D + j] = cos_t[(size_t)pos * 64 + j];
    cs[r * AB_D + 64 + j] = sin_t[(size_t)pos * 64 + j];
}

// NP: how the projection output arrives -- 0: a T tensor; > 0: that many fp32 split-K partial sums (compile-time, so that all
// loads of a batch are in flight together); -1: any number of partials (a loop; correct, slower)
template <typename TT, int NP>
__global__ __launch_bounds__(64 * AB_WAVES, 1) void k_attn_block(const typename TT::elem *__restrict__ qkv, int n_part_rt, long long part_stride,
                                                                const float *__restrict__ cs, typename TT::elem *__restrict__ kc,
                                                                typename TT::elem *__restrict__ vt, typename TT::elem *__restrict__ out,
                                                                int n_q_pad, int n_heads, int n_kv_heads, long long max_len,
                                                                const unsigned long long *__restrict__ mask, const int *__restrict__ d_Lw,
                                                                const int *__restrict__ d_Lvis, const int *__restrict__ d_n, float scale_log2) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    // ~122 KiB of LDS (one workgroup per CU): per-wave P rows | rotated Q rows | rotated K and plain V of the new rows | per-wave
    // partial O, m, l
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E *Pw = reinterpret_cast<E *>(smem);
    E *Qs = reinterpret_cast<E *>(smem + AB_LDS_QS);
    E *Kn = reinterpret_cast<E *>(smem + AB_LDS_KN);
    E *Vn = reinterpret_cast<E *>(smem + AB_LDS_VN);
    float *Om = reinterpret_cast<float *>(smem + AB_LDS_OM);
    float *Mm = reinterpret_cast<float *>(smem + AB_LDS_MM), *Lm = Mm + AB_WAVES * 16;

    const int h = blockIdx.x, rb = blockIdx.y;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lg = l >> 4;
    const int group = n_heads / n_kv_heads, kvh = h / group;
    const int W = (n_heads + 2 * n_kv_heads) * AB_D;                  // elements per row of the projection output
    const float *part = reinterpret_cast<const float *>(qkv);
    const bool fresh_wave = w == AB_WAVES - 1;
    const int nv = n_q_pad < AB_TILE ? n_q_pad : AB_TILE;             // rows of the projection output that may be new keys

    // 4 consecutive projection-output values at element offset off -> floats of T-rounded values
    auto ld4 = [&](size_t off, float (&x)[4]) {
        if constexpr (NP == 0) {
            const uint2 raw = *reinterpret_cast<const uint2 *>(qkv + off);
            const E *e = reinterpret_cast<const E *>(&raw);
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = (float)e[i];
        } else if constexpr (NP > 0) {
            float4 a[NP];
            const unsigned pst = (unsigned)part_stride;
#pragma unroll
            for (int k = 0; k < NP; k++) a[k] = *reinterpret_cast<const float4 *>(part + off + (unsigned)k * pst);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NP; k++) { acc[0] += a[k].x; acc[1] += a[k].y; acc[2] += a[k].z; acc[3] += a[k].w; }
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = (float)(E)acc[i];      // rounded like the GEMM's own output (k_rope_kv does the same)
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < n_part_rt; k++) {
                const float4 a = *reinterpret_cast<const float4 *>(part + (size_t)k * part_stride + off);
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = (float)(E)acc[i];
        }
    };
    // RoPE of elements e .. e + 4 and their partners 64 further on of projection row `row`, columns col0 + ..: HF rotate_half,
    // arithmetic identical to k_rope_kv; result to dst[e ..] and dst[e + 64 ..] (and, when kdst, to the K cache row)
    auto rope4 = [&](int row, int col0, int e, E *dst, bool to_cache, int Lw_, int n_) {
        float x1[4], x2[4];
        ld4((size_t)row * W + col0 + e, x1);
        ld4((size_t)row * W + col0 + e + 64, x2);
        const float4 c = *reinterpret_cast<const float4 *>(cs + row * AB_D + e), sn = *reinterpret_cast<const float4 *>(cs + row * AB_D + 64 + e);
        const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
        E lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo[i] = (E)(x1[i] * cc[i] - x2[i] * ss[i]);
            hi[i] = (E)(x2[i] * cc[i] + x1[i] * ss[i]);
        }
        *reinterpret_cast<uint2 *>(dst + e) = *reinterpret_cast<const uint2 *>(&lo[0]);
        *reinterpret_cast<uint2 *>(dst + e + 64) = *reinterpret_cast<const uint2 *>(&hi[0]);
        if (to_cache && row < n_ && (long long)Lw_ + row < max_len) {      // L and n are consumed only here, after the loads above went out
            E *kdst = kc + ((size_t)kvh * max_len + Lw_ + row) * AB_D;
            *reinterpret_cast<uint2 *>(kdst + e) = *reinterpret_cast<const uint2 *>(&lo[0]);
            *reinterpret_cast<uint2 *>(kdst + e + 64) = *reinterpret_cast<const uint2 *>(&hi[0]);
        }
    };

    // ---- batch 1 (needs neither L nor n): mask rows, this wave's first tile (waves 0..6)
    unsigned long long mrow[4];
#pragma unroll
    for (int r = 0; r < 4; r++) mrow[r] = mask[16 * rb + 4 * lg + r];
    const E *kbase = kc + (size_t)kvh * max_len * AB_D;
    const E *vbase = vt + (size_t)kvh * AB_D * max_len;
    uint4 kraw[4][4], vraw[8][2];
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            int key = key0 + 16 * st + lr;
            key = key < (int)max_len ? key : (int)max_len - 1;
            const E *kp = kbase + (size_t)key * AB_D + 8 * lg;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) kraw[st][kk] = *reinterpret_cast<const uint4 *>(kp + 32 * kk);
        }
        // V^T: B operand of PV for d block dt, key half kcx: lane (lr = d, lg) holds keys key0 + 32 kcx + 8 lg .. + 8 of column d
        const int kb = key0 + 8 * lg;
#pragma unroll
        for (int dt = 0; dt < 8; dt++) {
            const E *vp = vbase + (size_t)(16 * dt + lr) * max_len;
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++) {
                int k0 = kb + 32 * kcx;
                k0 = k0 + 8 <= (int)max_len ? k0 : (int)max_len - 8;          // max_len is a multiple of 8 (checked by the host)
                vraw[dt][kcx] = *reinterpret_cast<const uint4 *>(vp + k0);
            }
        }
    };
    if (!fresh_wave) load_tile(w * AB_TILE);
    const int Lw = d_Lw[0];
    const int Lvis = d_Lvis ? d_Lvis[0] : Lw;
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    const bool writer = (h % group) == 0 && rb == 0;

    // ---- all 512 threads: rotated Q rows of this block -> Qs; rotated K / plain V of the new rows -> Kn / Vn (+ the K cache rows)
    for (int u = tid; u < 16 * 16; u += 64 * AB_WAVES) {                  // unit = (row, 4 elements of the lower half + partners)
        const int r = u >> 4, e = 4 * (u & 15), qrow = 16 * rb + r;
        if (qrow < n_q_pad) rope4(qrow, h * AB_D, e, Qs + r * AB_QSTRIDE, false, 0, 0);
        else { *reinterpret_cast<uint2 *>(Qs + r * AB_QSTRIDE + e) = make_uint2(0, 0); *reinterpret_cast<uint2 *>(Qs + r * AB_QSTRIDE + e + 64) = make_uint2(0, 0); }
    }
    for (int u = 64 * AB_WAVES - 1 - tid; u < nv * 16; u += 64 * AB_WAVES) {    // reversed thread order: the Q units went to the low threads
        const int r = u >> 4, e = 4 * (u & 15);
        rope4(r, (n_heads + kvh) * AB_D, e, Kn + r * AB_QSTRIDE, writer, Lw, n);
    }
    for (int u = tid; u < nv * 32; u += 64 * AB_WAVES) {
        const int r = u >> 5, e = 4 * (u & 31);
        float x[4];
        ld4((size_t)r * W + (n_heads + n_kv_heads + kvh) * AB_D + e, x);
        E v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = r < n ? (E)x[i] : (E)0.f;       // rows that are not new keys read as zero
        *reinterpret_cast<uint2 *>(Vn + r * AB_QSTRIDE + e) = *reinterpret_cast<const uint2 *>(&v[0]);
    }
    __syncthreads();
    if (writer) {
        // V^T rows of the cache: column d gets the n new values at [Lw, Lw + n)
        for (int u = tid; u < AB_D * ((n + 3) / 4); u += 64 * AB_WAVES) {
            const int d = u % AB_D, j0 = 4 * (u / AB_D);
            E *dst = vt + ((size_t)kvh * AB_D + d) * max_len + Lw;
            for (int j = j0; j < j0 + 4 && j < n && (long long)Lw + j < max_len; j++) dst[j] = Vn[j * AB_QSTRIDE + d];
        }
    }
    uint4 qraw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) qraw[kk] = *reinterpret_cast<const uint4 *>(Qs + lr * AB_QSTRIDE + 32 * kk + 8 * lg);

    const int voff = Lw - Lvis;                            // mask bit of new row j is voff + j
    const int ntc = (Lw + AB_TILE - 1) / AB_TILE;          // tiles over the cached keys
#pragma unroll
    for (int r = 0; r < 4; r++) if (16 * rb + 4 * lg + r >= n) mrow[r] = 0ull;
    const bool rows_live = 16 * rb < n;                    // a row block beyond n only zeroes its output rows

    float m_run[4], l_run[4];
    floatx4 o[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    E *Pmine = Pw + w * 16 * AB_PSTRIDE;

    // one 64-key tile against this wave's 16 rows: keys below `vis_keys` (tile-relative) are visible to every row, key j >= vis_keys
    // is visible to row i iff bit (bit_base + j - vis_keys) of mask[i]; keys >= live are not keys at all (their V columns are
    // zeroed: 0 x stale bits could be NaN)
    auto tile_math = [&](int live, int vis_keys, int bit_base) {
        if (live < AB_TILE) {                               // wave-uniform: only a context's last tile and the new keys
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++) {
                const int k0 = 32 * kcx + 8 * lg;
                unsigned mlo[4];
#pragma unroll
                for (int d2 = 0; d2 < 4; d2++)
                    mlo[d2] = (k0 + 2 * d2 < live ? 0xFFFFu : 0u) | (k0 + 2 * d2 + 1 < live ? 0xFFFF0000u : 0u);
#pragma unroll
                for (int dt = 0; dt < 8; dt++) { vraw[dt][kcx].x &= mlo[0]; vraw[dt][kcx].y &= mlo[1]; vraw[dt][kcx].z &= mlo[2]; vraw[dt][kcx].w &= mlo[3]; }
            }
        }
        floatx4 s[4];
#pragma unroll
        for (int st = 0; st < 4; st++) {
            floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = TT::mfma(__builtin_bit_cast(V8, qraw[kk]), __builtin_bit_cast(V8, kraw[st][kk]), acc);
            s[st] = acc;
        }
        float tmax[4];
#pragma unroll
        for (int r = 0; r < 4; r++) tmax[r] = -INFINITY;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const int jl = 16 * st + lr;
            const int bit = jl >= live ? -2 : (jl < vis_keys ? -1 : bit_base + jl - vis_keys);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool ok = bit == -1 || (bit >= 0 && bit < 64 && ((mrow[r] >> bit) & 1ull));
                const float v = ok ? s[st][r] * scale_log2 : -INFINITY;
                s[st][r] = v;
                tmax[r] = fmaxf(tmax[r], v);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = row16_max(tmax[r]);          // DPP row reductions (samd_common.h): no ds_bpermute round trips in the tile loop
            const float m_new = fmaxf(m_run[r], v);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[r] - m_use);
            float psum = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const float p = __builtin_amdgcn_exp2f(s[st][r] - m_use);
                psum += p;
                Pmine[(4 * lg + r) * AB_PSTRIDE + 16 * st + lr] = (E)p;     // C layout -> A layout through this wave's own LDS rows
            }
            psum = row16_sum(psum);
            l_run[r] = l_run[r] * alpha + psum;
            m_run[r] = m_new;
#pragma unroll
            for (int dt = 0; dt < 8; dt++) o[dt][r] *= alpha;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // this wave's P stores before its P loads (one wave, in-order LDS)
        __builtin_amdgcn_wave_barrier();
        V8 pa[2];
#pragma unroll
        for (int kcx = 0; kcx < 2; kcx++)
            pa[kcx] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4 *>(&Pmine[lr * AB_PSTRIDE + 32 * kcx + 8 * lg]));
#pragma unroll
        for (int dt = 0; dt < 8; dt++)
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++) o[dt] = TT::mfma(pa[kcx], __builtin_bit_cast(V8, vraw[dt][kcx]), o[dt]);
        __builtin_amdgcn_wave_barrier();
    };

    if (rows_live) {
        // ---- wave 7 first: the n new keys as one tile out of Kn / Vn
        if (fresh_wave && n > 0) {
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int kk = 0; kk < 4; kk++)
                    kraw[st][kk] = 16 * st + lr < nv ? *reinterpret_cast<const uint4 *>(Kn + (16 * st + lr) * AB_QSTRIDE + 32 * kk + 8 * lg) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 8; dt++)
#pragma unroll
                for (int kcx = 0; kcx < 2; kcx++) {
                    unsigned short e[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int j = 32 * kcx + 8 * lg + i;
                        e[i] = j < nv ? *reinterpret_cast<const unsigned short *>(Vn + j * AB_QSTRIDE + 16 * dt + lr) : (unsigned short)0;
                    }
                    vraw[dt][kcx] = *reinterpret_cast<const uint4 *>(&e[0]);
                }
            tile_math(n, 0, voff);
        }
        // ---- cached tiles w, w + 8, ... (wave 7 loads its first one only now)
        for (int t = w; t < ntc; t += AB_WAVES) {
            const int key0 = t * AB_TILE;
            if (t != w || fresh_wave) load_tile(key0);
            const int live = Lw - key0 < AB_TILE ? Lw - key0 : AB_TILE;
            const int vis = Lvis - key0;                     // tile-relative first key that is governed by mask bits
            tile_math(live, vis < 0 ? 0 : (vis > AB_TILE ? AB_TILE : vis), vis < 0 ? -vis : 0);
        }
    }
    // ---- the 8 partials meet in LDS
    {
        float *om = Om + w * 16 * AB_OSTRIDE;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * lg + r;
#pragma unroll
            for (int dt = 0; dt < 8; dt++) om[row * AB_OSTRIDE + 16 * dt + lr] = o[dt][r];
            if (lr == 0) { Mm[w * 16 + row] = m_run[r]; Lm[w * 16 + row] = l_run[r]; }
        }
    }
    __syncthreads();
    {
        const int row = tid >> 5, d0 = 4 * (tid & 31), grow = 16 * rb + row;
        if (grow < n_q_pad) {
            float M = -INFINITY;
#pragma unroll
            for (int k = 0; k < AB_WAVES; k++) M = fmaxf(M, Mm[k * 16 + row]);
            float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
            for (int k = 0; k < AB_WAVES; k++) {
                const float mk = Mm[k * 16 + row];
                if (mk == -INFINITY) continue;
                const float wgt = exp2f(mk - M);
                den += wgt * Lm[k * 16 + row];
                const float4 v = *reinterpret_cast<const float4 *>(&Om[(k * 16 + row) * AB_OSTRIDE + d0]);
                num[0] += wgt * v.x; num[1] += wgt * v.y; num[2] += wgt * v.z; num[3] += wgt * v.w;
            }
            E e[4];
#pragma unroll
            for (int i = 0; i < 4; i++) e[i] = (E)((grow < n && den > 0.f) ? num[i] / den : 0.f);
            *reinterpret_cast<uint2 *>(out + ((size_t)grow * n_heads + h) * AB_D + d0) = *reinterpret_cast<const uint2 *>(&e[0]);
        }
    }
}

extern "C" {

int samd_rope_rows(const int32_t *d_rel_pos, const int32_t *d_base, const float *d_cos, const float *d_sin, float *d_cs, int32_t rows,
                   int32_t head_dim, int32_t max_pos, void *stream) {
    if (!d_rel_pos || !d_base || !d_cos || !d_sin || !d_cs || rows < 1 || rows > SAMD_MAX_DRAFT || head_dim != AB_D || max_pos < 1) {
        samd_set_error("samd_rope_rows: invalid argument (head_dim must be 128, rows <= 64)"); return SAMD_E_INVALID;
    }
    hipLaunchKernelGGL(k_rope_rows, dim3((rows * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rel_pos, d_base, d_cos, d_sin, d_cs, rows, max_pos);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_attention_block(const void *d_qkv, int32_t n_partials, int64_t partial_stride, const float *d_cs, void *d_k_cache, void *d_vt_cache,
                         void *d_out, int32_t dtype, int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                         const uint64_t *d_mask, const int32_t *d_write_pos, const int32_t *d_visible_len, const int32_t *d_n, float scale,
                         void *stream) {
    if (!d_qkv || !d_cs || !d_k_cache || !d_vt_cache || !d_out || !d_mask || !d_write_pos || !d_n) { samd_set_error("samd_attention_block: null pointer"); return SAMD_E_INVALID; }
    if (head_dim != AB_D || n_q_pad < 1 || n_q_pad > 64 || n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || n_partials < 0 ||
        max_len < 8 || max_len % 8 != 0 || (dtype != SAMD_F16 && dtype != SAMD_BF16)) {
        samd_set_error("samd_attention_block: unsupported shape (head_dim must be 128, n_q_pad <= 64, max_len a multiple of 8, f16/bf16)");
        return SAMD_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const float scale_log2 = scale * 1.4426950408889634f;
    const dim3 grid(n_heads, (n_q_pad + 15) / 16), block(64 * AB_WAVES);
#define GO(TT, NP, ET) do { static unsigned long long done_ = 0ull; \
        if (samd_reserve_lds((const void *)k_attn_block<TT, NP>, AB_LDS_BYTES, &done_) != hipSuccess) { samd_set_error("samd_attention_block: cannot reserve %d bytes of LDS on this device", (int)AB_LDS_BYTES); return SAMD_E_HIP; } \
        hipLaunchKernelGGL((k_attn_block<TT, NP>), grid, block, AB_LDS_BYTES, st, (const ET *)d_qkv, n_partials, (long long)partial_stride, d_cs, \
        (ET *)d_k_cache, (ET *)d_vt_cache, (ET *)d_out, n_q_pad, n_heads, n_kv_heads, (long long)max_len, (const unsigned long long *)d_mask, d_write_pos, \
        d_visible_len, d_n, scale_log2); } while (0)
#define GO2(TT, ET) do { if (n_partials == 0) GO(TT, 0, ET); else if (n_partials == 2) GO(TT, 2, ET); else if (n_partials == 5) GO(TT, 5, ET); else GO(TT, -1, ET); } while (0)
    if (dtype == SAMD_F16) GO2(AF16, _Float16); else GO2(ABF16, __bf16);
#undef GO2
#undef GO
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
