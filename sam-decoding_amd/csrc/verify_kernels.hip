// verify_kernels.hip -- gfx950 kernels of the verify side: row arg-max, KV-cache compaction,
// tree-mask attention (MFMA), Token-Recycle table.  Reference functions: see include/samd_hip.h.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "samd_common.h"
#include "topk_device.h"
#include "warm_device.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { samd_set_error("%s: %s", #x, hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct F16 { typedef _Float16 elem; typedef half8 vec8; };
struct BF16 { typedef __bf16 elem; typedef bf16x8 vec8; };

__device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(__bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(float x) { return x; }

// ================================================================================================
// torch.argmax(logits, -1): first maximum wins (SO/utils.py:86, :131)
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(1024) void k_argmax_rows(const T *__restrict__ logits, int rows, long long vocab, long long stride,
                                                      const int *__restrict__ d_rows, int *__restrict__ out) {
    const int row = blockIdx.x;
    if (d_rows && row >= d_rows[0]) return;
    if (row >= rows) return;
    const T *x = logits + (size_t)row * stride;
    float best = -INFINITY; long long bi = 0x7fffffffffffffffll;
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_ok = ((((size_t)x) & 15) == 0);
    const long long nvec = vec_ok ? vocab / VEC : 0;
    for (long long c = threadIdx.x; c < nvec; c += blockDim.x) {
        const uint4 raw = reinterpret_cast<const uint4 *>(x)[c];
        const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
        for (int j = 0; j < VEC; j++) { const float v = to_f32(e[j]); if (v > best) { best = v; bi = c * VEC + j; } }
    }
    for (long long i = nvec * VEC + threadIdx.x; i < vocab; i += blockDim.x) {
        const float v = to_f32(x[i]);
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    // reduce (value desc, index asc)
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const long long oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __shared__ float sv[16]; __shared__ long long si[16];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int k = 1; k < nw; k++) if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        out[row] = (bi == 0x7fffffffffffffffll) ? 0 : (int)bi;     // all -inf / NaN row: index 0
    }
}

// ================================================================================================
// SamdStaticCache.select_indices (SO/cache.py:118-133): gather accepted rows, then scatter them to
// [start, start+a).  One workgroup per (tensor, head): gather into LDS first (the reference's
// index_select materialises before copy_), so overlapping source/destination rows are safe.
// ================================================================================================
__global__ __launch_bounds__(256) void k_kv_compact(void *const *__restrict__ tensors, const int *__restrict__ verdict,
                                                    const int *__restrict__ kv_index, int n_heads, long long max_len, int row_bytes,
                                                    int h_start, int h_accept, int n_row_major) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int a = h_accept, start = h_start;
    if (verdict) {                                         // device-side verdict of the fused step
        if (!verdict[V_IS_TREE]) return;                   // sequence drafts keep their rows in place
        a = verdict[V_ACCEPT]; start = verdict[V_KV_START];
    }
    const int t = blockIdx.x / n_heads;
    unsigned char *base = (unsigned char *)tensors[t] + (size_t)(blockIdx.x % n_heads) * max_len * row_bytes;
    if (t >= n_row_major) {
        // a TRANSPOSED tensor [head][D][max_len] of 2-byte elements (the V cache of samd_attention_block): "row" start + idx[j] is
        // the column of every one of the D lines; gather all of them first, then scatter (same overlap rule as below)
        const int D = row_bytes >> 1, total = a * D;
        unsigned short *lds = reinterpret_cast<unsigned short *>(smem);
        unsigned short *col = reinterpret_cast<unsigned short *>(base);
        for (int c = threadIdx.x; c < total; c += blockDim.x) {
            const int d = c / a, j = c - d * a;
            lds[c] = col[(size_t)d * max_len + start + kv_index[j]];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < total; c += blockDim.x) {
            const int d = c / a, j = c - d * a;
            if (kv_index[j] == j) continue;
            col[(size_t)d * max_len + start + j] = lds[c];
        }
        return;
    }
    const int chunks = row_bytes >> 4, total = a * chunks;
    uint4 *lds = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int j = c / chunks, k = c - j * chunks;
        const long long src = (long long)start + kv_index[j];
        lds[c] = reinterpret_cast<const uint4 *>(base + (size_t)src * row_bytes)[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int j = c / chunks, k = c - j * chunks;
        if (kv_index[j] == j) continue;                    // already in place
        reinterpret_cast<uint4 *>(base + (size_t)(start + j) * row_bytes)[k] = lds[c];
    }
}

// ================================================================================================
// tree-mask attention (SO/model_patch/llama.py:82-96 + SDPA).  One workgroup = one query head x one
// KV split; 4 wavefronts x 16 query rows; 64-key tiles.  QK^T and PV on v_mfma_f32_16x16x32_{f16,bf16};
// K fragments straight from HBM/L2 (row-major K is already the B-operand layout), V staged through LDS
// transposed (key pairs packed per dword) so that PV's B fragments are 16-byte LDS reads; online softmax
// in registers in the exp2 domain; partial (m, l, O) per split, merged by k_attn_combine.
// ================================================================================================
#define ATT_TILE 64
#define ATT_SPLITS 16
#define ATT_D 128
#define VT_STRIDE (ATT_TILE + 8)      // halfs per Vt row: 144 B keeps 16-B alignment, spreads banks
#define P_STRIDE (ATT_TILE + 8)

template <typename TT> struct Mfma;
template <> struct Mfma<F16> {
    static __device__ __forceinline__ floatx4 run(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<BF16> {
    static __device__ __forceinline__ floatx4 run(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// blockIdx.x -> query head, XCD-aware (round 4).  Workgroups go to the 8 XCDs round-robin by linear id = blockIdx.x + gridDim.x * blockIdx.y,
// so with a head count that is a multiple of 8 all workgroups of one blockIdx.x share an XCD and its L2.  Heads are dealt to the XCDs in
// CONTIGUOUS groups of n_heads / 8: the query heads of one KV head (grouped-query attention: Llama-3 has 4 per KV head) then read that head's
// K / V through ONE L2 instead of four, and the merge launch (same mapping) finds a head's partials in the L2 that wrote them.
__device__ __forceinline__ int att_head_of_block(int bx, int n_heads) {
    return (n_heads & 7) == 0 ? (bx & 7) * (n_heads >> 3) + (bx >> 3) : bx;
}

// WIDE = a draft of 65..128 nodes: two 64-row tiles (blockIdx.z), two mask words per row.  The <= 64-node instantiation -- every step of every
// BASELINE configuration -- is the round-4 kernel again: one mask word, no second word to select (round 5 ran ONE generic kernel whose
// `rel < 64 ? lo >> rel : hi >> (rel - 64)` the compiler turned into exec-mask branches per score in the one computing wave's softmax:
// 8.81 -> 9.32 us per launch across the bench, +0.5 % on every row bucket's step)
// VT (round 6): the V half of the cache is TRANSPOSED ([H_kv][D][max_len], the layout samd_attention_block keeps): a tile's V^T image in LDS is then
// a straight copy -- four 16-byte loads + four 16-byte LDS stores per thread where the row-major cache needs sixteen 4-byte stores of repacked key
// pairs (4-way bank-conflicted) in front of the barrier every wave waits at.
template <typename TT, bool WIDE, bool VT = false>
__global__ __launch_bounds__(256) void k_tree_attention(const typename TT::elem *__restrict__ q, const typename TT::elem *__restrict__ kc,
                                                        const typename TT::elem *__restrict__ vc, float *__restrict__ ws,
                                                        int n_q_pad, int n_heads, int n_kv_heads, long long max_len,
                                                        const unsigned long long *__restrict__ mask, const int *__restrict__ d_L,
                                                        const int *__restrict__ d_n, float scale_log2, WarmArgs warm) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    __shared__ __attribute__((aligned(16))) E Vt[ATT_D * VT_STRIDE];
    __shared__ __attribute__((aligned(16))) E Pw[4 * 16 * P_STRIDE];

    const int h = att_head_of_block(blockIdx.x, n_heads), split = blockIdx.y;
    if (split >= ATT_SPLITS) {               // warm workgroups: the head of the output projection's weight stream -> this XCD's L2 (warm_device.h)
        if (blockIdx.z != 0) return;
        const unsigned a = warm_next_projection(warm, (split - ATT_SPLITS) * n_heads + h);
        if (a == 0x9E3779B9u && n_q_pad < 0) ws[0] = 0.f;         // never true: keeps the loads alive
        return;
    }
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lg = l >> 4;
    // blockIdx.z = the 64-row tile of the draft (round 5: drafts of up to 128 nodes are two tiles; row i's ancestor mask is two u64 words,
    // the high words -- nodes 64..127 -- stored SAMD_MAX_DRAFT entries behind the low ones; a draft of <= 64 nodes has one tile, low words only)
    const int row_base = (WIDE ? 64 * (int)blockIdx.z : 0) + 16 * w;
    // loads that do not depend on L / n go out first (Q fragments, mask rows), together with the two scalars
    V8 qa[4];
    {
        const int qrow = row_base + lr;
        const E *qp = q + ((size_t)qrow * n_heads + h) * ATT_D + 8 * lg;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (qrow < n_q_pad) raw = *reinterpret_cast<const uint4 *>(qp + 32 * kk);
            qa[kk] = __builtin_bit_cast(V8, raw);
        }
    }
    unsigned long long mrow[4], mrow_hi[WIDE ? 4 : 1];
#pragma unroll
    for (int r = 0; r < 4; r++) {                                             // the mask holds n_q_pad rows (<= 64: low words only); rows >= n are zeroed below
        mrow[r] = mask[row_base + 4 * lg + r];
        if constexpr (WIDE) mrow_hi[r] = mask[SAMD_MAX_DRAFT + row_base + 4 * lg + r];
    }
    // Split s owns the key tiles s, s + ATT_SPLITS, s + 2 ATT_SPLITS, ...: the FIRST tile of a workgroup is known from its
    // block index alone, so its K fragments and V rows are requested here, in the same memory round trip as the two scalars
    // (L, n) instead of after them.  Keys are clamped to the cache; what lies beyond L + n is masked (K) / zeroed (V) below.
    const int kvh = h / (n_heads / n_kv_heads);
    const E *kbase = kc + (size_t)kvh * max_len * ATT_D;
    const E *vbase = vc + (size_t)kvh * max_len * ATT_D;
    const bool may_be_active = row_base < n_q_pad;     // n <= n_q_pad: waves beyond the row bucket never compute
    uint4 kraw[4][4], vra[2], vrb[2];                  // (VT: vra[0..1], vrb[0..1] = this thread's four 16-byte pieces of the V^T tile)
    auto load_k = [&](int key0) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            int key = key0 + 16 * st + lr;
            key = key < (int)max_len ? key : (int)max_len - 1;
            const E *kp = kbase + (size_t)key * ATT_D + 8 * lg;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) kraw[st][kk] = *reinterpret_cast<const uint4 *>(kp + 32 * kk);
        }
    };
    auto load_v = [&](int key0) {                      // thread handles key pair (2p, 2p+1) x one 8-wide d chunk, twice
        if constexpr (VT) {                            // V^T: piece u = tid + 256 i: column d = u >> 3, keys key0 + 8 (u & 7) .. + 8 -- 128 contiguous bytes per d row
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int u = tid + 256 * i, d = u >> 3;
                int k0 = key0 + 8 * (u & 7);
                k0 = k0 + 8 <= (int)max_len ? k0 : (int)max_len - 8;              // max_len is a multiple of 8 (checked by the host); such keys lie beyond L + n
                const uint4 v = *reinterpret_cast<const uint4 *>(vbase + (size_t)d * max_len + k0);
                if (i == 0) vra[0] = v; else if (i == 1) vra[1] = v; else if (i == 2) vrb[0] = v; else vrb[1] = v;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int p = tid >> 3, ch = (tid & 7) + 8 * it, d0 = 8 * ch;      // a wave's instruction = 8 key rows x 128 contiguous bytes
            int ka = key0 + 2 * p, kb = ka + 1;
            ka = ka < (int)max_len ? ka : (int)max_len - 1; kb = kb < (int)max_len ? kb : (int)max_len - 1;
            vra[it] = *reinterpret_cast<const uint4 *>(vbase + (size_t)ka * ATT_D + d0);
            vrb[it] = *reinterpret_cast<const uint4 *>(vbase + (size_t)kb * ATT_D + d0);
        }
    };
    if (may_be_active) load_k(split * ATT_TILE);
    load_v(split * ATT_TILE);
    const int L = d_L[0];
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    const int total = L + n;
    const int ntiles = (total + ATT_TILE - 1) / ATT_TILE;
    if (split >= ntiles) return;                       // no keys for this split: k_attn_combine only reads the splits in use
    const bool active = row_base < n;
#pragma unroll
    for (int r = 0; r < 4; r++) if (row_base + 4 * lg + r >= n) { mrow[r] = 0ull; if constexpr (WIDE) mrow_hi[r] = 0ull; }

    float m_run[4], l_run[4];
    floatx4 o[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    E *Pmine = Pw + w * 16 * P_STRIDE;

    for (int t = split; t < ntiles; t += ATT_SPLITS) {
        const int key0 = t * ATT_TILE;
        // ---- later tiles: K fragment loads first, their latency overlaps the V staging below
        if (t != split && active) load_k(key0);
        __syncthreads();                                   // previous tile's Vt / Pw reads are done
        if (t != split) load_v(key0);
        // ---- stage V^T (key pairs packed per dword); rows past L + n are zero: their P is 0, but 0 x garbage could be NaN.
        //      A wave's load instruction covers 8 key rows x 128 contiguous bytes (round 4: 32 rows x 32 bytes before -- four times the
        //      requests for the same lines; 13.55 -> 13.25 us per layer at L = 800).  The LDS writes below are 4-way bank-conflicted in this
        //      order; an XOR swizzle of the key-pair index that removes the conflicts costs more in address arithmetic than it saves (13.5)
        if constexpr (VT) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int u = tid + 256 * i, d = u >> 3, kc8 = 8 * (u & 7);
                uint4 v = i == 0 ? vra[0] : (i == 1 ? vra[1] : (i == 2 ? vrb[0] : vrb[1]));
                const int live = total - key0 - kc8;                               // keys of this piece that exist (< 8: the context's last tile only)
                if (live < 8) {
                    unsigned m[4];
#pragma unroll
                    for (int d2 = 0; d2 < 4; d2++) m[d2] = (2 * d2 < live ? 0xFFFFu : 0u) | (2 * d2 + 1 < live ? 0xFFFF0000u : 0u);
                    v.x &= m[0]; v.y &= m[1]; v.z &= m[2]; v.w &= m[3];
                }
                *reinterpret_cast<uint4 *>(&Vt[d * VT_STRIDE + kc8]) = v;
            }
        } else
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int p = tid >> 3, ch = (tid & 7) + 8 * it, d0 = 8 * ch;      // a wave's instruction = 8 key rows x 128 contiguous bytes
            const int ka = key0 + 2 * p, kb = ka + 1;
            const uint4 ra = ka < total ? vra[it] : make_uint4(0, 0, 0, 0), rb = kb < total ? vrb[it] : make_uint4(0, 0, 0, 0);
            const unsigned short *ea = reinterpret_cast<const unsigned short *>(&ra);
            const unsigned short *eb = reinterpret_cast<const unsigned short *>(&rb);
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<unsigned int *>(&Vt[(d0 + j) * VT_STRIDE + 2 * p]) = (unsigned int)ea[j] | ((unsigned int)eb[j] << 16);
        }
        // ---- S = Q K^T for this wave's 16 rows x 64 keys
        floatx4 s[4];
        if (active) {
#pragma unroll
            for (int st = 0; st < 4; st++) {
                floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc = Mfma<TT>::run(qa[kk], __builtin_bit_cast(V8, kraw[st][kk]), acc);
                s[st] = acc;
            }
            // ---- mask, online softmax (exp2 domain); lane holds rows 4*lg+r, key 16*st+lr
            float tmax[4];
#pragma unroll
            for (int r = 0; r < 4; r++) tmax[r] = -INFINITY;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const int key = key0 + 16 * st + lr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rel = key - L;                                   // index of a NEW key among the draft's nodes
                    bool ok;
                    if constexpr (WIDE) {                                      // the word is SELECTED (v_cndmask), then one shift: no branch per score
                        const unsigned long long word = rel < 64 ? mrow[r] : mrow_hi[r];
                        ok = (key < L) | ((key < total) & (bool)((word >> (rel & 63)) & 1ull));
                    } else ok = (key < L) | ((key < total) & (bool)((mrow[r] >> (rel & 63)) & 1ull));        // bitwise: no exec-mask branch per score
                    const float v = ok ? s[st][r] * scale_log2 : -INFINITY;
                    s[st][r] = v;
                    tmax[r] = fmaxf(tmax[r], v);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = row16_max(tmax[r]);
                const float m_new = fmaxf(m_run[r], v);
                const float m_use = m_new == -INFINITY ? 0.f : m_new;
                // v_exp_f32 itself (arguments <= 0; results below 2^-126 flush to zero, as weights of that size may): exp2f() wraps it in a
                // denormal-range rescale -- a compare, two selects and an ldexp per score
                const float alpha = __builtin_amdgcn_exp2f(m_run[r] - m_use);          // m_run = -inf -> 0
                float psum = 0.f;
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    const float p = __builtin_amdgcn_exp2f(s[st][r] - m_use);
                    psum += p;
                    Pmine[(4 * lg + r) * P_STRIDE + 16 * st + lr] = (E)p;
                }
                psum = row16_sum(psum);
                l_run[r] = l_run[r] * alpha + psum;
                m_run[r] = m_new;
#pragma unroll
                for (int dt = 0; dt < 8; dt++) o[dt][r] *= alpha;
            }
        }
        __syncthreads();                                   // Vt staged, Pw written
        if (active) {
            // ---- O += P V : A = P[row=lr][keys 32*kc + 8*lg..+8], B = Vt[d = 16*dt + lr][same keys]
            V8 pa[2];
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++)
                pa[kcx] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4 *>(&Pmine[lr * P_STRIDE + 32 * kcx + 8 * lg]));
#pragma unroll
            for (int dt = 0; dt < 8; dt++) {
#pragma unroll
                for (int kcx = 0; kcx < 2; kcx++) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(&Vt[(16 * dt + lr) * VT_STRIDE + 32 * kcx + 8 * lg]);
                    o[dt] = Mfma<TT>::run(pa[kcx], __builtin_bit_cast(V8, raw), o[dt]);
                }
            }
        }
    }
    // ---- partial result: ws[split][row][h][0..127] = O, [128] = m, [129] = l
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row_base + 4 * lg + r;
            float *dst = ws + (((size_t)split * n_q_pad + row) * n_heads + h) * (ATT_D + 2);
            if (row < n_q_pad) {
#pragma unroll
                for (int dt = 0; dt < 8; dt++) dst[16 * dt + lr] = o[dt][r];
                if (lr == 0) { dst[ATT_D] = m_run[r]; dst[ATT_D + 1] = l_run[r]; }
            }
        }
    }
}

// <= 16 draft rows over a TRANSPOSED V cache (round 6): ONE wave per (head, split) -- the row-major kernel above keeps three more waves per
// workgroup whose only work is to repack the V tile into LDS in front of a barrier.  V^T in memory IS the B operand of P V (column d = 16 dt + lr,
// keys 32 kcx + 8 lg .. + 8: one 16-byte load per MFMA), so nothing is staged and no workgroup barrier exists; the only LDS traffic is the wave's
// own 16 x 64 P tile (C layout -> A layout).  Same MFMAs over the same key order as k_tree_attention: bit-identical partials.
template <typename TT>
__global__ __launch_bounds__(64) void k_tree_attention_direct(const typename TT::elem *__restrict__ q, const typename TT::elem *__restrict__ kc,
                                                               const typename TT::elem *__restrict__ vt, float *__restrict__ ws,
                                                               int n_q_pad, int n_heads, int n_kv_heads, long long max_len,
                                                               const unsigned long long *__restrict__ mask, const int *__restrict__ d_L,
                                                               const int *__restrict__ d_n, float scale_log2, WarmArgs warm) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    __shared__ __attribute__((aligned(16))) E Pw[16 * P_STRIDE];
    const int h = att_head_of_block(blockIdx.x, n_heads), split = blockIdx.y;
    if (split >= ATT_SPLITS) {
        if (blockIdx.z != 0) return;
        const unsigned a = warm_next_projection(warm, (split - ATT_SPLITS) * n_heads + h);
        if (a == 0x9E3779B9u && n_q_pad < 0) ws[0] = 0.f;
        return;
    }
    const int l = threadIdx.x, lr = l & 15, lg = l >> 4;
    const int row_base = 16 * (int)blockIdx.z;         // (experiment: wider drafts as independent 16-row blocks, each wave loading the tile itself)
    V8 qa[4];
    {
        const E *qp = q + ((size_t)(row_base + lr) * n_heads + h) * ATT_D + 8 * lg;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row_base + lr < n_q_pad) raw = *reinterpret_cast<const uint4 *>(qp + 32 * kk);
            qa[kk] = __builtin_bit_cast(V8, raw);
        }
    }
    unsigned long long mrow[4];
#pragma unroll
    for (int r = 0; r < 4; r++) mrow[r] = row_base + 4 * lg + r < n_q_pad ? mask[row_base + 4 * lg + r] : 0ull;
    const int kvh = h / (n_heads / n_kv_heads);
    const E *kbase = kc + (size_t)kvh * max_len * ATT_D;
    const E *vbase = vt + (size_t)kvh * max_len * ATT_D;
    uint4 kraw[4][4], vraw[8][2];
    auto load_kv = [&](int key0) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            int key = key0 + 16 * st + lr;
            key = key < (int)max_len ? key : (int)max_len - 1;
            const E *kp = kbase + (size_t)key * ATT_D + 8 * lg;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) kraw[st][kk] = *reinterpret_cast<const uint4 *>(kp + 32 * kk);
        }
#pragma unroll
        for (int kcx = 0; kcx < 2; kcx++) {
            int k0 = key0 + 32 * kcx + 8 * lg;
            k0 = k0 + 8 <= (int)max_len ? k0 : (int)max_len - 8;                  // max_len % 8 == 0 (host check); such keys lie beyond L + n
#pragma unroll
            for (int dt = 0; dt < 8; dt++) vraw[dt][kcx] = *reinterpret_cast<const uint4 *>(vbase + (size_t)(16 * dt + lr) * max_len + k0);
        }
    };
    load_kv(split * ATT_TILE);
    const int L = d_L[0];
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    const int total = L + n;
    const int ntiles = (total + ATT_TILE - 1) / ATT_TILE;
    if (split >= ntiles || row_base >= n) return;
#pragma unroll
    for (int r = 0; r < 4; r++) if (row_base + 4 * lg + r >= n) mrow[r] = 0ull;
    float m_run[4], l_run[4];
    floatx4 o[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    for (int t = split; t < ntiles; t += ATT_SPLITS) {
        const int key0 = t * ATT_TILE;
        if (t != split) load_kv(key0);
        floatx4 s[4];
#pragma unroll
        for (int st = 0; st < 4; st++) {
            floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = Mfma<TT>::run(qa[kk], __builtin_bit_cast(V8, kraw[st][kk]), acc);
            s[st] = acc;
        }
        float tmax[4];
#pragma unroll
        for (int r = 0; r < 4; r++) tmax[r] = -INFINITY;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const int key = key0 + 16 * st + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rel = key - L;
                const bool ok = (key < L) | ((key < total) & (bool)((mrow[r] >> (rel & 63)) & 1ull));
                const float v = ok ? s[st][r] * scale_log2 : -INFINITY;
                s[st][r] = v;
                tmax[r] = fmaxf(tmax[r], v);
            }
        }
        if (t != split) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // the previous tile's P reads are done
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = row16_max(tmax[r]);
            const float m_new = fmaxf(m_run[r], v);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[r] - m_use);
            float psum = 0.f;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const float p = __builtin_amdgcn_exp2f(s[st][r] - m_use);
                psum += p;
                Pw[(4 * lg + r) * P_STRIDE + 16 * st + lr] = (E)p;
            }
            psum = row16_sum(psum);
            l_run[r] = l_run[r] * alpha + psum;
            m_run[r] = m_new;
#pragma unroll
            for (int dt = 0; dt < 8; dt++) o[dt][r] *= alpha;
        }
        // keys past L + n: P is 0 there, but 0 x garbage could be NaN -- zero those V^T elements (the context's last tile only)
        if (key0 + ATT_TILE > total) {
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++) {
                const int live = total - key0 - 32 * kcx - 8 * lg;
                unsigned m[4];
#pragma unroll
                for (int d2 = 0; d2 < 4; d2++) m[d2] = (2 * d2 < live ? 0xFFFFu : 0u) | (2 * d2 + 1 < live ? 0xFFFF0000u : 0u);
#pragma unroll
                for (int dt = 0; dt < 8; dt++) { vraw[dt][kcx].x &= m[0]; vraw[dt][kcx].y &= m[1]; vraw[dt][kcx].z &= m[2]; vraw[dt][kcx].w &= m[3]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();                           // the wave's own P tile is written
        V8 pa[2];
#pragma unroll
        for (int kcx = 0; kcx < 2; kcx++) pa[kcx] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4 *>(&Pw[lr * P_STRIDE + 32 * kcx + 8 * lg]));
#pragma unroll
        for (int dt = 0; dt < 8; dt++) {
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++) o[dt] = Mfma<TT>::run(pa[kcx], __builtin_bit_cast(V8, vraw[dt][kcx]), o[dt]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = row_base + 4 * lg + r;
        float *dst = ws + (((size_t)split * n_q_pad + row) * n_heads + h) * (ATT_D + 2);
        if (row < n) {
#pragma unroll
            for (int dt = 0; dt < 8; dt++) dst[16 * dt + lr] = o[dt][r];
            if (lr == 0) { dst[ATT_D] = m_run[r]; dst[ATT_D + 1] = l_run[r]; }
        }
    }
}

template <typename E>
__global__ __launch_bounds__(128) void k_attn_combine(const float *__restrict__ ws, E *__restrict__ out, int n_q_pad, int n_heads,
                                                      const int *__restrict__ d_L, const int *__restrict__ d_n, WarmArgs warm, int *__restrict__ arrive) {
    // workgroup (head slot, row) with k_tree_attention's head mapping (att_head_of_block): a head's merge runs on the XCD whose L2 its sixteen
    // split workgroups just wrote the partials through (round 4; the grid was (row, head) before)
    const int h = att_head_of_block(blockIdx.x, n_heads), row = blockIdx.y, d = threadIdx.x;
    if (row >= n_q_pad) {                    // warm workgroups: the head of the output projection's weight stream -> this XCD's L2 (warm_device.h)
        const unsigned a = warm_next_projection(warm, (row - n_q_pad) * n_heads + h);
        if (a == 0x9E3779B9u && n_q_pad < 0) out[0] = (E)0.f;     // never true: keeps the loads alive
        return;
    }
    // every split's (m, l, O[d]) is loaded before anything is consumed, and before the two scalars that say how many
    // splits ran are known: one memory round trip for the whole kernel.  Splits that did not run hold older partials
    // (the workspace always has ATT_SPLITS slots); they are ignored below.
    // (m, l) of split s: ONE 8-byte load by lane s of each wave, handed round by shuffles -- as 32 wave-wide loads of one address each they
    // were two thirds of this launch's requests, and a latency-bound launch pays per request (profiles/r04_attention.md section 2d)
    float mv[ATT_SPLITS], lv[ATT_SPLITS], pv[ATT_SPLITS];
    const int lane = d & 63;
    float2 ml = make_float2(-INFINITY, 0.f);
    if (lane < ATT_SPLITS) ml = *reinterpret_cast<const float2 *>(ws + (((size_t)lane * n_q_pad + row) * n_heads + h) * (ATT_D + 2) + ATT_D);
#pragma unroll
    for (int s = 0; s < ATT_SPLITS; s++) pv[s] = ws[(((size_t)s * n_q_pad + row) * n_heads + h) * (ATT_D + 2) + d];
#pragma unroll
    for (int s = 0; s < ATT_SPLITS; s++) { mv[s] = __shfl(ml.x, s); lv[s] = __shfl(ml.y, s); }
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    E *dst = out + ((size_t)row * n_heads + h) * ATT_D + d;
    float res = 0.f;
    if (row < n) {
        // the splits k_tree_attention actually ran: split s owns the key tiles s, s + ATT_SPLITS, ...
        const int ntiles = (d_L[0] + n + ATT_TILE - 1) / ATT_TILE;
        const int used = ntiles < ATT_SPLITS ? ntiles : ATT_SPLITS;
        float M = -INFINITY;
#pragma unroll
        for (int s = 0; s < ATT_SPLITS; s++) if (s < used) M = fmaxf(M, mv[s]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int s = 0; s < ATT_SPLITS; s++) {
            if (s >= used || mv[s] == -INFINITY) continue;
            const float wgt = exp2f(mv[s] - M);
            num += wgt * pv[s]; den += wgt * lv[s];
        }
        res = den > 0.f ? num / den : 0.f;
    }
    *dst = (E)res;
    if (arrive) {
        // seam experiment (scripts/seam_probe.py, profiles/r04_attention.md section 5): a consumer on another queue polls this counter
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ================================================================================================
// k_tree_attention with RoPE and the K/V row write folded in (the k_rope_kv launch disappears): same 16 KV splits over the
// cached keys, plus ONE more workgroup per head (split ATT_SPLITS) that owns the n new keys.
//  * all 256 threads of a workgroup first rotate the block's Q rows straight out of the q|k|v projection's output (a T tensor or
//    the streaming GEMM's fp32 split-K partials) into LDS -- their loads go out in the same round trip as the first K/V tile and
//    the two scalars; cos/sin per row come from k_rope_rows (once per forward), so nothing here waits for L;
//  * the workgroup of the new keys is known from the block index, so it requests the new rows' k and v at entry as well, rotates
//    k into LDS (and, for the first query head of a KV group, into the cache) and keeps v in the registers the V^T staging reads;
//  * cached tiles never see rows >= L (masked / zeroed), so nobody reads what that workgroup is writing;
//  * partial (m, l, O) per slot; k_attn_combine_slots merges the ATT_SPLITS + 1 slots (a second launch: the in-kernel merge by
//    the last-arriving workgroup measured slower than the launch it saves, profiles/r02_attention_variants.md).
// ================================================================================================
#define ATT_SLOTS (ATT_SPLITS + 1)
#define QS_STRIDE (ATT_D + 8)         // halfs per staged Q / new-K row: 272 B, 16 rows of one ds_read_b128 hit different banks

template <typename TT, int NP>
__global__ __launch_bounds__(256, 2) void k_tree_attention_rope(const typename TT::elem *__restrict__ qkv, int n_part_rt, long long part_stride,
                                                               const float *__restrict__ cs, typename TT::elem *__restrict__ kc,
                                                               typename TT::elem *__restrict__ vc, float *__restrict__ ws,
                                                               int n_q_pad, int n_heads, int n_kv_heads, long long max_len,
                                                               const unsigned long long *__restrict__ mask, const int *__restrict__ d_L,
                                                               const int *__restrict__ d_n, float scale_log2) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    __shared__ __attribute__((aligned(16))) E Vt[ATT_D * VT_STRIDE];
    __shared__ __attribute__((aligned(16))) E Pw[4 * 16 * P_STRIDE];
    __shared__ __attribute__((aligned(16))) E Qs[ATT_TILE * QS_STRIDE];       // rotated Q rows; the fresh split reuses it for the new keys' K after its Q fragments are read

    const int h = blockIdx.x, split = blockIdx.y;
    const bool fresh = split == ATT_SPLITS;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lg = l >> 4;
    const int row_base = 16 * w;
    const int group = n_heads / n_kv_heads, kvh = h / group;
    const int W = (n_heads + 2 * n_kv_heads) * ATT_D;
    const float *part = reinterpret_cast<const float *>(qkv);

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
    // up to 4 rope units of this thread, all loads first: unit u = (row r = u / 16, elements e = 4 (u % 16) .. + 4 and their partners
    // 64 further on) of the projection rows [0, rows) at column col0; rotated values -> dst rows (LDS) and, for live rows of the
    // writer, to the K cache at [L + r]  (HF rotate_half; arithmetic identical to k_rope_kv)
    auto rope_units = [&](int rows, int col0, E *dst, bool to_cache, const int *pL, const int *pn) {
        float x1[4][4], x2[4][4];
        float4 c[4], sn[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = tid + 256 * k;
            if (u >= rows * 16) continue;                          // wave-uniform for k >= 1 (rows is a multiple of... any count: guarded again below)
            const int r = u >> 4, e = 4 * (u & 15);
            ld4((size_t)r * W + col0 + e, x1[k]);
            ld4((size_t)r * W + col0 + e + 64, x2[k]);
            c[k] = *reinterpret_cast<const float4 *>(cs + r * ATT_D + e);
            sn[k] = *reinterpret_cast<const float4 *>(cs + r * ATT_D + 64 + e);
        }
        const int Lw = to_cache ? pL[0] : 0, nn = to_cache ? pn[0] : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = tid + 256 * k;
            if (u >= rows * 16) continue;
            const int r = u >> 4, e = 4 * (u & 15);
            const float cc[4] = {c[k].x, c[k].y, c[k].z, c[k].w}, ss[4] = {sn[k].x, sn[k].y, sn[k].z, sn[k].w};
            E lo[4], hi[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                lo[i] = (E)(x1[k][i] * cc[i] - x2[k][i] * ss[i]);
                hi[i] = (E)(x2[k][i] * cc[i] + x1[k][i] * ss[i]);
            }
            *reinterpret_cast<uint2 *>(dst + r * QS_STRIDE + e) = *reinterpret_cast<const uint2 *>(&lo[0]);
            *reinterpret_cast<uint2 *>(dst + r * QS_STRIDE + e + 64) = *reinterpret_cast<const uint2 *>(&hi[0]);
            if (to_cache && r < nn && (long long)Lw + r < max_len) {
                E *kdst = kc + ((size_t)kvh * max_len + Lw + r) * ATT_D;
                *reinterpret_cast<uint2 *>(kdst + e) = *reinterpret_cast<const uint2 *>(&lo[0]);
                *reinterpret_cast<uint2 *>(kdst + e + 64) = *reinterpret_cast<const uint2 *>(&hi[0]);
            }
        }
    };

    // ---- loads that need neither L nor n: mask rows, this workgroup's first tile (cached splits) or the new rows' v (fresh split)
    unsigned long long mrow[4];
#pragma unroll
    for (int r = 0; r < 4; r++) mrow[r] = mask[row_base + 4 * lg + r];
    const E *kbase = kc + (size_t)kvh * max_len * ATT_D;
    const E *vbase = vc + (size_t)kvh * max_len * ATT_D;
    const bool may_be_active = row_base < n_q_pad;
    uint4 kraw[4][4], vra[2], vrb[2];
    auto load_k = [&](int key0) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            int key = key0 + 16 * st + lr;
            key = key < (int)max_len ? key : (int)max_len - 1;
            const E *kp = kbase + (size_t)key * ATT_D + 8 * lg;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) kraw[st][kk] = *reinterpret_cast<const uint4 *>(kp + 32 * kk);
        }
    };
    auto load_v = [&](int key0) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int p = tid & 31, ch = (tid >> 5) + 8 * it, d0 = 8 * ch;
            int ka = key0 + 2 * p, kb = ka + 1;
            ka = ka < (int)max_len ? ka : (int)max_len - 1; kb = kb < (int)max_len ? kb : (int)max_len - 1;
            vra[it] = *reinterpret_cast<const uint4 *>(vbase + (size_t)ka * ATT_D + d0);
            vrb[it] = *reinterpret_cast<const uint4 *>(vbase + (size_t)kb * ATT_D + d0);
        }
    };
    const bool writer = (h % group) == 0;
    if (!fresh) {
        if (may_be_active) load_k(split * ATT_TILE);
        load_v(split * ATT_TILE);
    } else {
        // the new rows' v: thread (p, ch) holds rows 2p, 2p + 1 x columns 8 ch .. + 8, exactly what the V^T staging below consumes
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int p = tid & 31, ch = (tid >> 5) + 8 * it, d0 = 8 * ch;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int j = 2 * p + half;
                uint4 raw = make_uint4(0, 0, 0, 0);
                if (j < n_q_pad) {
                    float xa[4], xb[4];
                    const size_t off = (size_t)j * W + (n_heads + n_kv_heads + kvh) * ATT_D + d0;
                    ld4(off, xa); ld4(off + 4, xb);
                    E e[8];
#pragma unroll
                    for (int i = 0; i < 4; i++) { e[i] = (E)xa[i]; e[4 + i] = (E)xb[i]; }
                    raw = *reinterpret_cast<const uint4 *>(&e[0]);
                }
                if (half == 0) vra[it] = raw; else vrb[it] = raw;
            }
        }
    }
    // ---- rotated Q rows of all n_q_pad rows -> Qs (every workgroup needs the Q of its head)
    rope_units(n_q_pad, h * ATT_D, Qs, false, nullptr, nullptr);
    const int L = d_L[0];
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    const int ntc = (L + ATT_TILE - 1) / ATT_TILE;
    const bool has_fresh = n > 0;
    const int used_c = ntc < 1 ? (has_fresh ? 0 : 1) : (ntc < ATT_SPLITS ? ntc : ATT_SPLITS);   // with nothing to do split 0 still writes an (empty) slot
    __syncthreads();                                       // Qs complete
    uint4 qraw[4];
    {
        const int qrow = row_base + lr;
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            qraw[kk] = qrow < n_q_pad ? *reinterpret_cast<const uint4 *>(Qs + qrow * QS_STRIDE + 32 * kk + 8 * lg) : make_uint4(0, 0, 0, 0);
    }
    if (fresh ? !has_fresh : split >= used_c) return;
    const bool active = row_base < n;
#pragma unroll
    for (int r = 0; r < 4; r++) if (row_base + 4 * lg + r >= n) mrow[r] = 0ull;
    if (fresh) {
        __syncthreads();                                   // every wave has its Q fragments: Qs becomes the new keys' K
        rope_units(n_q_pad, (n_heads + kvh) * ATT_D, Qs, writer, d_L, d_n);
        if (writer) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int p = tid & 31, ch = (tid >> 5) + 8 * it, d0 = 8 * ch;
                if (2 * p < n && (long long)L + 2 * p < max_len) *reinterpret_cast<uint4 *>(vc + ((size_t)kvh * max_len + L + 2 * p) * ATT_D + d0) = vra[it];
                if (2 * p + 1 < n && (long long)L + 2 * p + 1 < max_len) *reinterpret_cast<uint4 *>(vc + ((size_t)kvh * max_len + L + 2 * p + 1) * ATT_D + d0) = vrb[it];
            }
        }
        __syncthreads();                                   // the new keys' K rows are in Qs
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                kraw[st][kk] = 16 * st + lr < n_q_pad ? *reinterpret_cast<const uint4 *>(Qs + (16 * st + lr) * QS_STRIDE + 32 * kk + 8 * lg) : make_uint4(0, 0, 0, 0);
    }

    float m_run[4], l_run[4];
    floatx4 o[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    E *Pmine = Pw + w * 16 * P_STRIDE;

    const int t_end = fresh ? 1 : ntc;
    for (int t = fresh ? 0 : split; t < t_end; t += ATT_SPLITS) {
        const int key0 = fresh ? L : t * ATT_TILE;
        if (!fresh && t != split && active) load_k(key0);
        __syncthreads();                                   // previous tile's Vt / Pw reads are done
        if (!fresh && t != split) load_v(key0);
        // ---- stage V^T (key pairs packed per dword); keys beyond the tile's live range are zero: their P is 0, but 0 x garbage could be NaN
        const int live = fresh ? n : (L - key0 < ATT_TILE ? L - key0 : ATT_TILE);
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int p = tid & 31, ch = (tid >> 5) + 8 * it, d0 = 8 * ch;
            const uint4 ra = 2 * p < live ? vra[it] : make_uint4(0, 0, 0, 0), rb = 2 * p + 1 < live ? vrb[it] : make_uint4(0, 0, 0, 0);
            const unsigned short *ea = reinterpret_cast<const unsigned short *>(&ra);
            const unsigned short *eb = reinterpret_cast<const unsigned short *>(&rb);
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<unsigned int *>(&Vt[(d0 + j) * VT_STRIDE + 2 * p]) = (unsigned int)ea[j] | ((unsigned int)eb[j] << 16);
        }
        floatx4 s[4];
        if (active) {
#pragma unroll
            for (int st = 0; st < 4; st++) {
                floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc = Mfma<TT>::run(__builtin_bit_cast(V8, qraw[kk]), __builtin_bit_cast(V8, kraw[st][kk]), acc);
                s[st] = acc;
            }
            float tmax[4];
#pragma unroll
            for (int r = 0; r < 4; r++) tmax[r] = -INFINITY;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const int jl = 16 * st + lr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool ok = jl < live && (!fresh || ((mrow[r] >> jl) & 1ull));      // cached keys < L are visible to every row
                    const float v = ok ? s[st][r] * scale_log2 : -INFINITY;
                    s[st][r] = v;
                    tmax[r] = fmaxf(tmax[r], v);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = row16_max(tmax[r]);
                const float m_new = fmaxf(m_run[r], v);
                const float m_use = m_new == -INFINITY ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[r] - m_use);
                float psum = 0.f;
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    const float p = __builtin_amdgcn_exp2f(s[st][r] - m_use);
                    psum += p;
                    Pmine[(4 * lg + r) * P_STRIDE + 16 * st + lr] = (E)p;
                }
                psum = row16_sum(psum);
                l_run[r] = l_run[r] * alpha + psum;
                m_run[r] = m_new;
#pragma unroll
                for (int dt = 0; dt < 8; dt++) o[dt][r] *= alpha;
            }
        }
        __syncthreads();                                   // Vt staged, Pw written
        if (active) {
            V8 pa[2];
#pragma unroll
            for (int kcx = 0; kcx < 2; kcx++)
                pa[kcx] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4 *>(&Pmine[lr * P_STRIDE + 32 * kcx + 8 * lg]));
#pragma unroll
            for (int dt = 0; dt < 8; dt++) {
#pragma unroll
                for (int kcx = 0; kcx < 2; kcx++) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(&Vt[(16 * dt + lr) * VT_STRIDE + 32 * kcx + 8 * lg]);
                    o[dt] = Mfma<TT>::run(pa[kcx], __builtin_bit_cast(V8, raw), o[dt]);
                }
            }
        }
    }
    // ---- partial result: ws[slot][row][h][0..127] = O, [128] = m, [129] = l
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row_base + 4 * lg + r;
            float *dst = ws + (((size_t)split * n_q_pad + row) * n_heads + h) * (ATT_D + 2);
            if (row < n_q_pad) {
#pragma unroll
                for (int dt = 0; dt < 8; dt++) dst[16 * dt + lr] = o[dt][r];
                if (lr == 0) { dst[ATT_D] = m_run[r]; dst[ATT_D + 1] = l_run[r]; }
            }
        }
    }
}

// merge of the ATT_SLOTS partials: one workgroup per (row, head), every slot's (m, l, O[d]) requested before anything is consumed
template <typename E>
__global__ __launch_bounds__(128) void k_attn_combine_slots(const float *__restrict__ ws, E *__restrict__ out, int n_q_pad, int n_heads,
                                                            const int *__restrict__ d_L, const int *__restrict__ d_n) {
    const int row = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    float mv[ATT_SLOTS], lv[ATT_SLOTS], pv[ATT_SLOTS];
#pragma unroll
    for (int s = 0; s < ATT_SLOTS; s++) {
        const float *p = ws + (((size_t)s * n_q_pad + row) * n_heads + h) * (ATT_D + 2);
        mv[s] = p[ATT_D]; lv[s] = p[ATT_D + 1]; pv[s] = p[d];
    }
    int n = d_n[0]; n = n > n_q_pad ? n_q_pad : n;
    E *dst = out + ((size_t)row * n_heads + h) * ATT_D + d;
    if (row >= n) { *dst = (E)0.f; return; }
    const int ntc = (d_L[0] + ATT_TILE - 1) / ATT_TILE;
    const int used_c = ntc < ATT_SPLITS ? ntc : ATT_SPLITS;        // n > 0 here, so the fresh slot is live
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < ATT_SLOTS; s++) if (s < used_c || s == ATT_SPLITS) M = fmaxf(M, mv[s]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < ATT_SLOTS; s++) {
        if (!(s < used_c || s == ATT_SPLITS) || mv[s] == -INFINITY) continue;
        const float wgt = exp2f(mv[s] - M);
        num += wgt * pv[s]; den += wgt * lv[s];
    }
    *dst = (E)(den > 0.f ? num / den : 0.f);
}

// ================================================================================================
// Token Recycle (S/tree_model/token_recycle/token_recycle.py:33-60)
// ================================================================================================
struct samd_recycle {
    int32_t vocab, n_nodes, n_levels, tmp_rows;
    int32_t *d_table;       // [vocab][8]
    uint8_t *d_present;     // [vocab]
    int32_t *d_tmp;         // [tmp_rows][8] top-8 of the rows of one update
    float *d_part_v; int32_t *d_part_i; int32_t n_split;     // [tmp_rows][n_split][8] candidates of the split top-8 (n_split = 0: one workgroup per row)
    int32_t *d_child_off, *d_children, *d_level_off, *d_level_nodes;
};

// logits.topk(8).indices per row in (value desc, index asc) order -- torch.topk's order on tie-free rows.
// One pass over the row: every thread keeps the best 8 of its strided elements in registers (insertion into a sorted
// list), the 256 x 8 survivors meet in LDS and 8 block arg-max rounds pick the winners.
__device__ __forceinline__ bool topk_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

template <typename T>
__global__ __launch_bounds__(256) void k_topk8_rows(const T *__restrict__ logits, int rows, long long vocab, long long stride,
                                                    const int *__restrict__ d_rows, int *__restrict__ out) {
    const int row = blockIdx.x;
    if ((d_rows && row >= d_rows[0]) || row >= rows) return;
    const T *x = logits + (size_t)row * stride;
    float bv[8]; int bi[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { bv[k] = -INFINITY; bi[k] = 0x7fffffff; }
    for (long long i = threadIdx.x; i < vocab; i += blockDim.x) {
        float v = to_f32(x[i]); int id = (int)i;
        if (!topk_before(v, id, bv[7], bi[7])) continue;
#pragma unroll
        for (int k = 0; k < 8; k++) {                      // sorted insertion: swap the carried element down the list
            if (topk_before(v, id, bv[k], bi[k])) { const float tv = bv[k]; const int ti = bi[k]; bv[k] = v; bi[k] = id; v = tv; id = ti; }
        }
    }
    __shared__ float sv[256 * 8]; __shared__ int si[256 * 8];
    __shared__ float wv[4]; __shared__ int wi[4], wslot[4];
#pragma unroll
    for (int k = 0; k < 8; k++) { sv[threadIdx.x * 8 + k] = bv[k]; si[threadIdx.x * 8 + k] = bi[k]; }
    __syncthreads();
    for (int round = 0; round < 8; round++) {
        float best = -INFINITY; int bidx = 0x7fffffff, bslot = -1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int slot = threadIdx.x * 8 + k;
            if (si[slot] != 0x7fffffff && (bslot < 0 || topk_before(sv[slot], si[slot], best, bidx))) { best = sv[slot]; bidx = si[slot]; bslot = slot; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bidx, o), os = __shfl_xor(bslot, o);
            if (os >= 0 && (bslot < 0 || topk_before(ov, oi, best, bidx))) { best = ov; bidx = oi; bslot = os; }
        }
        if ((threadIdx.x & 63) == 0) { wv[threadIdx.x >> 6] = best; wi[threadIdx.x >> 6] = bidx; wslot[threadIdx.x >> 6] = bslot; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; k++)
                if (wslot[k] >= 0 && (bslot < 0 || topk_before(wv[k], wi[k], best, bidx))) { best = wv[k]; bidx = wi[k]; bslot = wslot[k]; }
            out[(size_t)row * 8 + round] = bslot < 0 ? 0 : bidx;
            if (bslot >= 0) si[bslot] = 0x7fffffff;          // consumed
        }
        __syncthreads();
    }
}

// The same result from rows x ceil(vocab / 4096) workgroups + one merge workgroup per row (topk_device.h; the form EAGLE-2's
// rowstats use): the one-workgroup-per-row kernel above is VALU-bound on its CU -- scalar 2-byte loads, and with 64 lanes per wave
// some lane inserts into its sorted list at almost every element: 118 us for the 61 rows of a Token-Recycle step at a 32 k
// vocabulary, against 12 us for this form.
template <typename T>
__global__ __launch_bounds__(256) void k_topk8_part(const T *__restrict__ logits, int rows, long long vocab, long long stride, const int *__restrict__ d_rows,
                                                    int n_split, float *__restrict__ part_v, int *__restrict__ part_i) {
    const int row = blockIdx.x, split = blockIdx.y;
    if ((d_rows && row >= d_rows[0]) || row >= rows) return;
    __shared__ float sv[4 * E2_K]; __shared__ int si[4 * E2_K];
    float v[E2_EPT]; int id[E2_EPT];
    e2_load_segment<T>(logits + (size_t)row * stride, vocab, (long long)split * E2_SEG, v, id);
    float res_v[E2_K]; int res_i[E2_K];
    e2_block_top8<E2_EPT>(v, id, res_v, res_i, sv, si);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < E2_K; k++) { part_v[((size_t)row * n_split + split) * E2_K + k] = res_v[k]; part_i[((size_t)row * n_split + split) * E2_K + k] = res_i[k]; }
    }
}

__global__ __launch_bounds__(256) void k_topk8_merge(int rows, const int *__restrict__ d_rows, int n_split, const float *__restrict__ part_v,
                                                     const int *__restrict__ part_i, int *__restrict__ out) {
    const int row = blockIdx.x, tid = threadIdx.x;
    if ((d_rows && row >= d_rows[0]) || row >= rows) return;
    __shared__ float sv[4 * E2_K]; __shared__ int si[4 * E2_K];
    float v[2]; int id[2];
    const int n_cand = n_split * E2_K;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int c = tid + 256 * q;
        v[q] = c < n_cand ? part_v[(size_t)row * n_cand + c] : -INFINITY;
        id[q] = c < n_cand ? part_i[(size_t)row * n_cand + c] : 0x7fffffff;
        if (id[q] == 0x7fffffff) v[q] = -INFINITY;
    }
    float out_v[E2_K]; int out_i[E2_K];
    e2_block_top8<2>(v, id, out_v, out_i, sv, si);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < E2_K; k++) out[(size_t)row * 8 + k] = out_i[k] == 0x7fffffff ? 0 : out_i[k];
    }
}

// cache[token] = top-8, rows in order, later rows overwrite earlier ones (token_recycle.py:47-48)
__global__ void k_recycle_scatter(const int *__restrict__ tokens, const int *__restrict__ topk, int rows, const int *__restrict__ d_rows,
                                  int *__restrict__ table, uint8_t *__restrict__ present, int vocab) {
    const int n = d_rows ? min(rows, d_rows[0]) : rows;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = tokens[i];
    if (t < 0 || t >= vocab) return;
    for (int j = i + 1; j < n; j++) if (tokens[j] == t) return;        // a later row wins
    for (int k = 0; k < 8; k++) table[(size_t)t * 8 + k] = topk[(size_t)i * 8 + k];
    present[t] = 1;
}

// gen_draft (token_recycle.py:50-60): level-synchronous fill of the static tree, one lane per node
__global__ __launch_bounds__(64) void k_recycle_draft(const int *__restrict__ table, const uint8_t *__restrict__ present, int vocab,
                                                      const int *__restrict__ child_off, const int *__restrict__ children,
                                                      const int *__restrict__ level_off, const int *__restrict__ level_nodes,
                                                      int n_nodes, int n_levels, const int *__restrict__ start, int *__restrict__ out) {
    __shared__ int tok[SAMD_MAX_DRAFT];
    const int lane = threadIdx.x;
    for (int k = lane; k < n_nodes; k += 64) tok[k] = k == 0 ? start[0] : 0;
    __syncthreads();
    for (int lv = 0; lv < n_levels; lv++) {
        for (int k = level_off[lv] + lane; k < level_off[lv + 1]; k += 64) {
            const int node = level_nodes[k], t = tok[node];
            if (t >= 0 && t < vocab && present[t])
                for (int c = child_off[node]; c < child_off[node + 1]; c++) tok[children[c]] = table[(size_t)t * 8 + (c - child_off[node])];
        }
        __syncthreads();
    }
    for (int k = lane; k < n_nodes; k += 64) out[k] = tok[k];
}

// ================================================================================================
// C ABI
// ================================================================================================

// ---- eval_posterior, sampling branch (reference: samd_sam_only/utils.py:142-184) ---------------------------------------------
// The reference walks the candidate rows depth by depth on the host: at depth i the rows that still carry the accepted prefix
// offer their distinct next tokens in row order; token x is accepted with probability p(x) under the warped distribution of the
// accepted node (one host random.random() per examined token, utils.py:165); a rejection zeroes p(x) and renormalises
// (gtp = gtp / gtp.sum(), in the logits' dtype).  Here: probs [C * D][V] = softmax of the warped logits of every (row, position)
// (the caller runs HF's warpers over the whole batch on the device), uniforms = the next values of the host's random stream
// in order; one workgroup does the walk -- thread 0 the scalar trie logic, all threads the renormalisation -- and reports
// out = {best row, accept length, uniforms consumed, 1 when `work` holds the residual distribution the reference returns as
// sample_p (utils.py:176-177), status (1 = ran out of uniforms)}.  Rounding follows torch on a tensor of dtype T: the sum is
// accumulated in fp32 and rounded to T, every quotient is rounded to T, r is compared in T.
template <typename T>
__global__ __launch_bounds__(1024) void k_posterior_sampled(const T *__restrict__ probs, const long long *__restrict__ cand, int C, int D, long long V,
                                                            const double *__restrict__ uniforms, int n_uniforms, T *__restrict__ work, int *__restrict__ out,
                                                            const int *__restrict__ rowmap, int n_rows) {
    __shared__ long long prefix[SAMD_MAX_DRAFT];
    __shared__ int seen[1024];
    __shared__ int s_action, s_tok, s_row, s_anchor;
    __shared__ float red[16];
    const int tid = threadIdx.x;
    int n_acc = 1, best = 0, k = 0, adjusted = 0, status = 0;
    if (tid == 0) prefix[0] = cand[0];
    __syncthreads();
    while (n_acc < D && !status) {
        adjusted = 0;
        // the first row that carries the accepted prefix: its (row, n_acc - 1) distribution is the accepted node's
        if (tid == 0) {
            int a = -1;
            for (int j = 0; j < C && a < 0; j++) {
                bool eq = true;
                for (int q = 0; q < n_acc; q++) eq = eq && cand[(size_t)j * D + q] == prefix[q];
                if (eq) a = j;
            }
            s_anchor = a;
        }
        __syncthreads();
        const int anchor = s_anchor;
        if (anchor < 0) break;
        // rowmap: probs holds one row per DRAFT NODE and cell (row, position) reads node rowmap[row * D + position]; -1 = the last
        // node (the reference gathers logits[retrieve] with -1 entries, samd_model.py:144).  Without it: one row per cell.
        size_t prow = (size_t)anchor * D + (n_acc - 1);
        if (rowmap) { const int m = rowmap[prow]; prow = (size_t)(m < 0 || m >= n_rows ? n_rows - 1 : m); }
        const T *src = probs + prow * V;
        for (long long i = tid; i < V; i += blockDim.x) work[i] = src[i];
        __syncthreads();
        int n_seen = 0, row = 0;
        bool grown = false;
        while (true) {
            if (tid == 0) {                                  // next row that offers a token not tried at this depth
                int action = 0, tok = -1, hit = -1;
                while (row < C) {
                    const int j = row++;
                    bool eq = true;
                    for (int q = 0; q < n_acc; q++) eq = eq && cand[(size_t)j * D + q] == prefix[q];
                    if (!eq) continue;
                    const long long x = cand[(size_t)j * D + n_acc];
                    if (x < 0) continue;                     // -1 = padding of a shorter candidate (utils.py:162)
                    bool dup = false;
                    for (int q = 0; q < n_seen && q < 1024; q++) dup = dup || seen[q] == (int)x;
                    if (dup) continue;
                    if (n_seen < 1024) seen[n_seen] = (int)x;
                    n_seen++;
                    if (k >= n_uniforms) { action = 3; break; }
                    const T r = (T)(float)uniforms[k++];     // torch compares the host scalar in the tensor's dtype
                    tok = (int)x; hit = j;
                    action = (float)r <= (float)work[tok] ? 1 : 2;
                    if (action == 1) prefix[n_acc] = x;
                    break;
                }
                s_action = action; s_tok = tok; s_row = hit;
            }
            __syncthreads();
            const int action = s_action, tok = s_tok;
            if (action == 0) break;                          // no more rows at this depth
            if (action == 3) { status = 1; break; }
            if (action == 1) { best = s_row; n_acc++; grown = true; break; }
            // rejected: zero the token's mass and renormalise
            if (tid == 0) work[tok] = (T)0.f;
            __syncthreads();
            float acc = 0.f;
            for (long long i = tid; i < V; i += blockDim.x) acc += (float)work[i];
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) tot += red[w];
            const float denom = (float)(T)tot;
            for (long long i = tid; i < V; i += blockDim.x) work[i] = (T)((float)work[i] / denom);
            adjusted = 1;
            __syncthreads();
        }
        __syncthreads();
        if (!grown) break;
    }
    if (tid == 0) { out[0] = best; out[1] = n_acc; out[2] = k; out[3] = (adjusted && n_acc != D) ? 1 : 0; out[4] = status; }
}

extern "C" {

int samd_argmax_rows(const void *d_logits, int32_t dtype, int32_t rows, int64_t vocab, int64_t row_stride, const int32_t *d_rows,
                     int32_t *d_out, void *stream) {
    if (!d_logits || !d_out || rows < 0 || vocab < 1 || row_stride < vocab) { samd_set_error("samd_argmax_rows: invalid argument"); return SAMD_E_INVALID; }
    if (rows == 0) return SAMD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == SAMD_F32 ? 4 : 2;
    if ((row_stride * esz) % 16 != 0 && rows > 1) {
        // rows after the first are not 16-byte aligned: the kernel falls back to scalar loads per row
    }
    switch (dtype) {
    case SAMD_F16: hipLaunchKernelGGL(k_argmax_rows<_Float16>, dim3(rows), dim3(1024), 0, st, (const _Float16 *)d_logits, rows, (long long)vocab, (long long)row_stride, d_rows, d_out); break;
    case SAMD_BF16: hipLaunchKernelGGL(k_argmax_rows<__bf16>, dim3(rows), dim3(1024), 0, st, (const __bf16 *)d_logits, rows, (long long)vocab, (long long)row_stride, d_rows, d_out); break;
    case SAMD_F32: hipLaunchKernelGGL(k_argmax_rows<float>, dim3(rows), dim3(1024), 0, st, (const float *)d_logits, rows, (long long)vocab, (long long)row_stride, d_rows, d_out); break;
    default: samd_set_error("samd_argmax_rows: bad dtype"); return SAMD_E_INVALID;
    }
    LAUNCHCHK();
    return SAMD_OK;
}

static int kv_compact(samd_session_t *s, void *const *d_tensors, int32_t n_tensors, int32_t n_transposed, int32_t n_heads, int64_t max_len,
                      int32_t head_dim, int32_t elem_bytes, int32_t start, const int32_t *d_indices, int32_t accept, void *stream) {
    if (!d_tensors || n_tensors < 1 || n_transposed < 0 || n_transposed > n_tensors || n_heads < 1 || head_dim < 1 || (head_dim * elem_bytes) % 16 != 0 ||
        (n_transposed > 0 && elem_bytes != 2)) {
        samd_set_error("samd_kv_compact: invalid argument (row bytes must be a multiple of 16; transposed tensors hold 2-byte elements)"); return SAMD_E_INVALID;
    }
    const int row_bytes = head_dim * elem_bytes;
    const size_t lds = (size_t)SAMD_MAX_DRAFT * row_bytes;
    if (lds > 64 * 1024) { samd_set_error("samd_kv_compact: row too large"); return SAMD_E_INVALID; }
    if (s) hipLaunchKernelGGL(k_kv_compact, dim3(n_tensors * n_heads), dim3(256), lds, (hipStream_t)stream, d_tensors, s->dev.verdict, s->dev.kv_index, n_heads,
                              (long long)max_len, row_bytes, 0, 0, n_tensors - n_transposed);
    else hipLaunchKernelGGL(k_kv_compact, dim3(n_tensors * n_heads), dim3(256), lds, (hipStream_t)stream, d_tensors, (const int *)nullptr, d_indices, n_heads,
                            (long long)max_len, row_bytes, start, accept, n_tensors - n_transposed);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_posterior_sampled(const void *d_probs, int32_t dtype, const int64_t *d_candidates, int32_t n_candidates, int32_t depth, int64_t vocab,
                           const double *d_uniforms, int32_t n_uniforms, void *d_work, int32_t *d_out, void *stream) {
    return samd_posterior_sampled_nodes(d_probs, dtype, nullptr, 0, d_candidates, n_candidates, depth, vocab, d_uniforms, n_uniforms, d_work, d_out, stream);
}

int samd_posterior_sampled_nodes(const void *d_probs, int32_t dtype, const int32_t *d_rowmap, int32_t n_rows, const int64_t *d_candidates, int32_t n_candidates,
                                 int32_t depth, int64_t vocab, const double *d_uniforms, int32_t n_uniforms, void *d_work, int32_t *d_out, void *stream) {
    if (!d_probs || !d_candidates || n_candidates < 1 || depth < 1 || depth > SAMD_MAX_DRAFT || vocab < 1 || !d_uniforms || n_uniforms < 0 || !d_work || !d_out ||
        (d_rowmap && n_rows < 1)) {
        samd_set_error("samd_posterior_sampled: invalid argument (depth <= 128)"); return SAMD_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAMD_F32) hipLaunchKernelGGL(k_posterior_sampled<float>, dim3(1), dim3(1024), 0, st, (const float *)d_probs, (const long long *)d_candidates, n_candidates, depth,
                                              (long long)vocab, d_uniforms, n_uniforms, (float *)d_work, d_out, d_rowmap, n_rows);
    else if (dtype == SAMD_F16) hipLaunchKernelGGL(k_posterior_sampled<_Float16>, dim3(1), dim3(1024), 0, st, (const _Float16 *)d_probs, (const long long *)d_candidates, n_candidates,
                                                   depth, (long long)vocab, d_uniforms, n_uniforms, (_Float16 *)d_work, d_out, d_rowmap, n_rows);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_posterior_sampled<__bf16>, dim3(1), dim3(1024), 0, st, (const __bf16 *)d_probs, (const long long *)d_candidates, n_candidates,
                                                    depth, (long long)vocab, d_uniforms, n_uniforms, (__bf16 *)d_work, d_out, d_rowmap, n_rows);
    else { samd_set_error("samd_posterior_sampled: bad dtype"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_kv_compact(samd_session_t *s, void *const *d_tensors, int32_t n_tensors, int32_t n_heads, int64_t max_len,
                    int32_t head_dim, int32_t elem_bytes, void *stream) {
    if (!s) { samd_set_error("samd_kv_compact: null session"); return SAMD_E_INVALID; }
    return kv_compact(s, d_tensors, n_tensors, 0, n_heads, max_len, head_dim, elem_bytes, 0, nullptr, 0, stream);
}

int samd_kv_compact_indices(void *const *d_tensors, int32_t n_tensors, int32_t n_heads, int64_t max_len, int32_t head_dim,
                            int32_t elem_bytes, int32_t start, const int32_t *d_indices, int32_t accept, void *stream) {
    if (!d_indices || start < 0 || accept < 0 || accept > SAMD_MAX_DRAFT || start + accept > max_len) { samd_set_error("samd_kv_compact_indices: invalid argument"); return SAMD_E_INVALID; }
    if (accept == 0) return SAMD_OK;
    return kv_compact(nullptr, d_tensors, n_tensors, 0, n_heads, max_len, head_dim, elem_bytes, start, d_indices, accept, stream);
}

// the same over a pointer table whose LAST n_transposed tensors are transposed ([head][D][max_len], 2-byte elements): the K tensors
// followed by the V^T tensors of a cache that samd_attention_block reads
int samd_kv_compact_vt(samd_session_t *s, void *const *d_tensors, int32_t n_tensors, int32_t n_transposed, int32_t n_heads, int64_t max_len,
                       int32_t head_dim, int32_t elem_bytes, void *stream) {
    if (!s) { samd_set_error("samd_kv_compact_vt: null session"); return SAMD_E_INVALID; }
    return kv_compact(s, d_tensors, n_tensors, n_transposed, n_heads, max_len, head_dim, elem_bytes, 0, nullptr, 0, stream);
}

int samd_kv_compact_indices_vt(void *const *d_tensors, int32_t n_tensors, int32_t n_transposed, int32_t n_heads, int64_t max_len, int32_t head_dim,
                               int32_t elem_bytes, int32_t start, const int32_t *d_indices, int32_t accept, void *stream) {
    if (!d_indices || start < 0 || accept < 0 || accept > SAMD_MAX_DRAFT || start + accept > max_len) { samd_set_error("samd_kv_compact_indices_vt: invalid argument"); return SAMD_E_INVALID; }
    if (accept == 0) return SAMD_OK;
    return kv_compact(nullptr, d_tensors, n_tensors, n_transposed, n_heads, max_len, head_dim, elem_bytes, start, d_indices, accept, stream);
}

int64_t samd_tree_attention_workspace(int32_t n_q_pad, int32_t n_heads, int32_t head_dim) {
    return (int64_t)ATT_SPLITS * n_q_pad * n_heads * (head_dim + 2) * 4;
}

int samd_tree_attention(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                        int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                        const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                        void *stream) {
    return samd_tree_attention_warm(d_q, d_k_cache, d_v_cache, d_out, dtype, n_q_pad, n_heads, n_kv_heads, head_dim, max_len, d_mask, d_cache_length, d_n,
                                    scale, d_workspace, workspace_bytes, nullptr, stream);
}

static int att_direct_rows() {                 // SAMD_ATT_DIRECT_ROWS: the widest row bucket that takes the one-wave kernel over a transposed-V cache (0: none; A/B switch)
    static const int rows = [] { const char *e = getenv("SAMD_ATT_DIRECT_ROWS"); return e ? atoi(e) : 16; }();
    return rows;
}

static int tree_attention_impl(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                               const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                               const samd_warm_t *next, int *d_arrive, void *stream, int v_transposed = 0) {
    if (v_transposed && (max_len < 8 || max_len % 8 != 0)) { samd_set_error("samd_tree_attention_vt: max_len must be a multiple of 8"); return SAMD_E_INVALID; }
    // the output projection's warm-up rides on the split launch (the longest glue launch of a layer) or on the merge launch
    const WarmArgs none = warm_args(nullptr);
    const WarmArgs wa_split = next && next->where == 0 ? warm_args(next) : none, wa = next && next->where != 0 ? warm_args(next) : none;
    const int warm_rows = (warm_blocks(wa) + n_heads - 1) / n_heads;                         // extra blockIdx.y rows of the merge launch
    const int warm_splits = (warm_blocks(wa_split) + n_heads - 1) / n_heads;                 // extra blockIdx.y rows of the split launch
    if (!d_q || !d_k_cache || !d_v_cache || !d_out || !d_mask || !d_cache_length || !d_n || !d_workspace) { samd_set_error("samd_tree_attention: null pointer"); return SAMD_E_INVALID; }
    if (head_dim != ATT_D || n_q_pad < 1 || n_q_pad > SAMD_MAX_DRAFT || n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 ||
        (dtype != SAMD_F16 && dtype != SAMD_BF16) || workspace_bytes < samd_tree_attention_workspace(n_q_pad, n_heads, head_dim)) {
        samd_set_error("samd_tree_attention: unsupported shape (head_dim must be 128, n_q_pad <= 128, f16/bf16) or workspace too small");
        return SAMD_E_INVALID;
    }
    const int row_tiles = (n_q_pad + 63) / 64;                 // > 64 rows: d_mask holds [2][SAMD_MAX_DRAFT] words (low, high)
    hipStream_t st = (hipStream_t)stream;
    const float scale_log2 = scale * 1.4426950408889634f;
    float *ws = (float *)d_workspace;
#define ATT_GO(TT, ET, W) if (v_transposed && n_q_pad <= att_direct_rows()) \
                          hipLaunchKernelGGL((k_tree_attention_direct<TT>), dim3(n_heads, ATT_SPLITS + warm_splits, (n_q_pad + 15) / 16), dim3(64), 0, st, (const ET *)d_q, \
                           (const ET *)d_k_cache, (const ET *)d_v_cache, ws, n_q_pad, n_heads, n_kv_heads, (long long)max_len, \
                           (const unsigned long long *)d_mask, d_cache_length, d_n, scale_log2, wa_split); \
                      else if (v_transposed) hipLaunchKernelGGL((k_tree_attention<TT, W, true>), dim3(n_heads, ATT_SPLITS + warm_splits, row_tiles), dim3(256), 0, st, (const ET *)d_q, \
                           (const ET *)d_k_cache, (const ET *)d_v_cache, ws, n_q_pad, n_heads, n_kv_heads, (long long)max_len, \
                           (const unsigned long long *)d_mask, d_cache_length, d_n, scale_log2, wa_split); \
                      else hipLaunchKernelGGL((k_tree_attention<TT, W>), dim3(n_heads, ATT_SPLITS + warm_splits, row_tiles), dim3(256), 0, st, (const ET *)d_q, \
                           (const ET *)d_k_cache, (const ET *)d_v_cache, ws, n_q_pad, n_heads, n_kv_heads, (long long)max_len, \
                           (const unsigned long long *)d_mask, d_cache_length, d_n, scale_log2, wa_split)
    if (dtype == SAMD_F16) {
        if (row_tiles > 1) { ATT_GO(F16, _Float16, true); } else { ATT_GO(F16, _Float16, false); }
        hipLaunchKernelGGL(k_attn_combine<_Float16>, dim3(n_heads, n_q_pad + warm_rows), dim3(ATT_D), 0, st, ws, (_Float16 *)d_out, n_q_pad, n_heads, d_cache_length, d_n, wa, d_arrive);
    } else {
        if (row_tiles > 1) { ATT_GO(BF16, __bf16, true); } else { ATT_GO(BF16, __bf16, false); }
        hipLaunchKernelGGL(k_attn_combine<__bf16>, dim3(n_heads, n_q_pad + warm_rows), dim3(ATT_D), 0, st, ws, (__bf16 *)d_out, n_q_pad, n_heads, d_cache_length, d_n, wa, d_arrive);
    }
#undef ATT_GO
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_tree_attention_warm(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                             int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                             const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                             const samd_warm_t *next, void *stream) {
    return tree_attention_impl(d_q, d_k_cache, d_v_cache, d_out, dtype, n_q_pad, n_heads, n_kv_heads, head_dim, max_len, d_mask, d_cache_length, d_n, scale,
                               d_workspace, workspace_bytes, next, nullptr, stream);
}

/* round 6: the same attention (samd_tree_attention_warm: `next` may be NULL) over a cache whose V half is TRANSPOSED ([H_kv][D][max_len], as
 * samd_attention_block keeps it; samd_gemm_qkv_rope*_vt, samd_rope_kv_write*_vt and samd_kv_compact*_vt maintain it): at <= 16 rows ONE wave per (head, split) feeds
 * its MFMAs straight from the V^T rows (k_tree_attention_direct: no LDS staging, no barrier); wider drafts copy the V^T tile into LDS instead of repacking it. */
int samd_tree_attention_vt(const void *d_q, const void *d_k_cache, const void *d_vt_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                           int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                           const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                           const samd_warm_t *next, void *stream) {
    return tree_attention_impl(d_q, d_k_cache, d_vt_cache, d_out, dtype, n_q_pad, n_heads, n_kv_heads, head_dim, max_len, d_mask, d_cache_length, d_n, scale,
                               d_workspace, workspace_bytes, next, nullptr, stream, 1);
}

int samd_tree_attention_signal(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                               const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                               int32_t *d_arrive, void *stream) {
    if (!d_arrive) return SAMD_E_INVALID;
    return tree_attention_impl(d_q, d_k_cache, d_v_cache, d_out, dtype, n_q_pad, n_heads, n_kv_heads, head_dim, max_len, d_mask, d_cache_length, d_n, scale,
                               d_workspace, workspace_bytes, nullptr, d_arrive, stream);
}

int64_t samd_tree_attention_rope_workspace(int32_t n_q_pad, int32_t n_heads, int32_t head_dim) {
    return (int64_t)ATT_SLOTS * n_q_pad * n_heads * (head_dim + 2) * 4;
}

// samd_rope_kv_write + samd_tree_attention in two launches instead of three: d_qkv is the q|k|v projection's output (dtype rows, or
// n_partials fp32 partial sums of partial_stride elements each), d_cs the per-row cos|sin of samd_rope_rows; K / V rows of the n new
// keys are written to the (row-major) caches at [L, L + n)
int samd_tree_attention_rope(const void *d_qkv, int32_t n_partials, int64_t partial_stride, const float *d_cs, void *d_k_cache, void *d_v_cache,
                             void *d_out, int32_t dtype, int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                             const uint64_t *d_mask, const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace,
                             int64_t workspace_bytes, void *stream) {
    if (!d_qkv || !d_cs || !d_k_cache || !d_v_cache || !d_out || !d_mask || !d_cache_length || !d_n || !d_workspace) { samd_set_error("samd_tree_attention_rope: null pointer"); return SAMD_E_INVALID; }
    if (head_dim != ATT_D || n_q_pad < 1 || n_q_pad > 64 || n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || n_partials < 0 ||
        (dtype != SAMD_F16 && dtype != SAMD_BF16) || workspace_bytes < samd_tree_attention_rope_workspace(n_q_pad, n_heads, head_dim)) {
        samd_set_error("samd_tree_attention_rope: unsupported shape (head_dim must be 128, n_q_pad <= 64, f16/bf16) or workspace too small");
        return SAMD_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const float scale_log2 = scale * 1.4426950408889634f;
    float *ws = (float *)d_workspace;
#define GO(TT, NP, ET) hipLaunchKernelGGL((k_tree_attention_rope<TT, NP>), dim3(n_heads, ATT_SLOTS), dim3(256), 0, st, (const ET *)d_qkv, n_partials, (long long)partial_stride, \
        d_cs, (ET *)d_k_cache, (ET *)d_v_cache, ws, n_q_pad, n_heads, n_kv_heads, (long long)max_len, (const unsigned long long *)d_mask, d_cache_length, d_n, scale_log2)
#define GO2(TT, ET) do { if (n_partials == 0) GO(TT, 0, ET); else if (n_partials == 2) GO(TT, 2, ET); else if (n_partials == 5) GO(TT, 5, ET); else GO(TT, -1, ET); \
        hipLaunchKernelGGL(k_attn_combine_slots<ET>, dim3(n_q_pad, n_heads), dim3(ATT_D), 0, st, ws, (ET *)d_out, n_q_pad, n_heads, d_cache_length, d_n); } while (0)
    if (dtype == SAMD_F16) GO2(F16, _Float16); else GO2(BF16, __bf16);
#undef GO2
#undef GO
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_recycle_create(int32_t vocab, const int32_t *h_child_offsets, const int32_t *h_children, int32_t n_nodes, samd_recycle_t **out) {
    if (!out || vocab < 1 || n_nodes < 1 || n_nodes > SAMD_MAX_DRAFT || !h_child_offsets) { samd_set_error("samd_recycle_create: invalid argument"); return SAMD_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { samd_set_error("no HIP device"); return SAMD_E_NODEVICE; }
    // levels of the static tree (parents precede children in the reference's numbering)
    std::vector<int> parent(n_nodes, -1), depth(n_nodes, 0);
    const int n_child = h_child_offsets[n_nodes];
    for (int i = 0; i < n_nodes; i++)
        for (int c = h_child_offsets[i]; c < h_child_offsets[i + 1]; c++) {
            const int ch = h_children[c];
            if (ch <= i || ch >= n_nodes || (c - h_child_offsets[i]) >= SAMD_TOPK) { samd_set_error("samd_recycle_create: bad tree"); return SAMD_E_INVALID; }
            parent[ch] = i;
        }
    int n_levels = 1;
    for (int i = 1; i < n_nodes; i++) { depth[i] = parent[i] < 0 ? 0 : depth[parent[i]] + 1; n_levels = std::max(n_levels, depth[i] + 1); }
    std::vector<int> level_off(n_levels + 1, 0), level_nodes(n_nodes);
    for (int i = 0; i < n_nodes; i++) level_off[depth[i] + 1]++;
    for (int lv = 0; lv < n_levels; lv++) level_off[lv + 1] += level_off[lv];
    { std::vector<int> fill(level_off.begin(), level_off.end() - 1); for (int i = 0; i < n_nodes; i++) level_nodes[fill[depth[i]]++] = i; }

    samd_recycle_t *t = (samd_recycle_t *)calloc(1, sizeof(samd_recycle_t));
    if (!t) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    t->vocab = vocab; t->n_nodes = n_nodes; t->n_levels = n_levels; t->tmp_rows = 16384;
    bool ok = hipMalloc((void **)&t->d_table, (size_t)vocab * 8 * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&t->d_present, (size_t)vocab) == hipSuccess;
    ok = ok && hipMalloc((void **)&t->d_tmp, (size_t)t->tmp_rows * 8 * 4) == hipSuccess;
    t->n_split = (vocab + E2_SEG - 1) / E2_SEG;
    if (t->n_split > E2_MAXSPLIT) t->n_split = 0;
    if (t->n_split) {
        ok = ok && hipMalloc((void **)&t->d_part_v, (size_t)t->tmp_rows * t->n_split * E2_K * 4) == hipSuccess;
        ok = ok && hipMalloc((void **)&t->d_part_i, (size_t)t->tmp_rows * t->n_split * E2_K * 4) == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&t->d_child_off, (n_nodes + 1) * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&t->d_children, std::max(1, n_child) * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&t->d_level_off, (n_levels + 1) * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&t->d_level_nodes, n_nodes * 4) == hipSuccess;
    ok = ok && hipMemset(t->d_table, 0, (size_t)vocab * 8 * 4) == hipSuccess && hipMemset(t->d_present, 0, (size_t)vocab) == hipSuccess;
    ok = ok && hipMemcpy(t->d_child_off, h_child_offsets, (n_nodes + 1) * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (n_child) ok = ok && hipMemcpy(t->d_children, h_children, n_child * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->d_level_off, level_off.data(), (n_levels + 1) * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->d_level_nodes, level_nodes.data(), n_nodes * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { samd_recycle_free(t); samd_set_error("samd_recycle_create: device allocation failed"); return SAMD_E_HIP; }
    *out = t;
    return SAMD_OK;
}

void samd_recycle_free(samd_recycle_t *t) {
    if (!t) return;
    void *ptrs[] = { t->d_table, t->d_present, t->d_tmp, t->d_part_v, t->d_part_i, t->d_child_off, t->d_children, t->d_level_off, t->d_level_nodes };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    free(t);
}

int samd_recycle_update(samd_recycle_t *t, const int32_t *d_tokens, const void *d_logits, int32_t dtype, int32_t n, const int32_t *d_n,
                        int64_t vocab, int64_t row_stride, void *stream) {
    if (!t || !d_tokens || !d_logits || n < 0 || n > t->tmp_rows || vocab < 8 || vocab > t->vocab) { samd_set_error("samd_recycle_update: invalid argument"); return SAMD_E_INVALID; }
    if (n == 0) return SAMD_OK;
    hipStream_t st = (hipStream_t)stream;
    // the split form needs the segments of THIS call's vocabulary to fit the table's workspace (vocab <= t->vocab holds)
    const int n_split = t->n_split ? (int)((vocab + E2_SEG - 1) / E2_SEG) : 0;
    if (n_split && (dtype == SAMD_F16 || dtype == SAMD_BF16)) {
        const dim3 grid(n, n_split);
        if (dtype == SAMD_F16) hipLaunchKernelGGL(k_topk8_part<_Float16>, grid, dim3(256), 0, st, (const _Float16 *)d_logits, n, (long long)vocab, (long long)row_stride, d_n, n_split, t->d_part_v, t->d_part_i);
        else hipLaunchKernelGGL(k_topk8_part<__bf16>, grid, dim3(256), 0, st, (const __bf16 *)d_logits, n, (long long)vocab, (long long)row_stride, d_n, n_split, t->d_part_v, t->d_part_i);
        hipLaunchKernelGGL(k_topk8_merge, dim3(n), dim3(256), 0, st, n, d_n, n_split, (const float *)t->d_part_v, (const int *)t->d_part_i, t->d_tmp);
    } else
    switch (dtype) {
    case SAMD_F16: hipLaunchKernelGGL(k_topk8_rows<_Float16>, dim3(n), dim3(256), 0, st, (const _Float16 *)d_logits, n, (long long)vocab, (long long)row_stride, d_n, t->d_tmp); break;
    case SAMD_BF16: hipLaunchKernelGGL(k_topk8_rows<__bf16>, dim3(n), dim3(256), 0, st, (const __bf16 *)d_logits, n, (long long)vocab, (long long)row_stride, d_n, t->d_tmp); break;
    case SAMD_F32: hipLaunchKernelGGL(k_topk8_rows<float>, dim3(n), dim3(256), 0, st, (const float *)d_logits, n, (long long)vocab, (long long)row_stride, d_n, t->d_tmp); break;
    default: return SAMD_E_INVALID;
    }
    hipLaunchKernelGGL(k_recycle_scatter, dim3((n + 255) / 256), dim3(256), 0, st, d_tokens, t->d_tmp, n, d_n, t->d_table, t->d_present, t->vocab);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_recycle_draft(samd_recycle_t *t, const int32_t *d_start_token, int32_t *d_out, void *stream) {
    if (!t || !d_start_token || !d_out) return SAMD_E_INVALID;
    hipLaunchKernelGGL(k_recycle_draft, dim3(1), dim3(64), 0, (hipStream_t)stream, t->d_table, t->d_present, t->vocab, t->d_child_off,
                       t->d_children, t->d_level_off, t->d_level_nodes, t->n_nodes, t->n_levels, d_start_token, d_out);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_recycle_export(samd_recycle_t *t, int32_t *h_table, uint8_t *h_present, void *stream) {
    if (!t) return SAMD_E_INVALID;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (h_table) HIPCHK(hipMemcpy(h_table, t->d_table, (size_t)t->vocab * 8 * 4, hipMemcpyDeviceToHost));
    if (h_present) HIPCHK(hipMemcpy(h_present, t->d_present, (size_t)t->vocab, hipMemcpyDeviceToHost));
    return SAMD_OK;
}

}  // extern "C"
