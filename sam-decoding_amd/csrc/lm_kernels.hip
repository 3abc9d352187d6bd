// lm_kernels.hip -- the memory-bound glue of the verify forward between the library GEMMs:
// token embedding gather, RMSNorm (+ fused residual add), RoPE + KV-cache row write at a device-side
// offset, SiLU*up.  They take every dynamic scalar (cache length L, draft size n, positions) from
// device memory so that a whole decode step is capturable in one hipGraph without host round trips.
// Reference semantics: HF LlamaDecoderLayer as driven by samd_sam_only/samd_model.py:134-138 with
// SamdStaticCache.update (samd_sam_only/cache.py:103-115) writing K/V at [cache_length, +n).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "samd_common.h"
#include "prefill_attn_device.h"
#include "warm_device.h"

#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

template <typename T> struct Vec8 { T v[8]; };

template <typename T> __device__ __forceinline__ Vec8<T> ld8(const T *p) {
    const uint4 raw = *reinterpret_cast<const uint4 *>(p);
    return __builtin_bit_cast(Vec8<T>, raw);
}
template <typename T> __device__ __forceinline__ void st8(T *p, const Vec8<T> &x) {
    *reinterpret_cast<uint4 *>(p) = __builtin_bit_cast(uint4, x);
}

// 8 consecutive values of a GEMM output that exists either as a T tensor (n_part == 0) or as fp32 split-K partial sums
// [n_part][...] (samd_gemm_skinny); the sum is rounded to T, as the GEMM's own epilogue would have done
template <typename T> __device__ __forceinline__ Vec8<T> ld8_or_partials(const T *src, const float *part, int n_part, size_t part_stride,
                                                                        size_t off) {
    if (n_part == 0) return ld8(src + off);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    // groups of 8 splits with every load issued before any add: one memory round trip per group instead of one per split
    for (int s0 = 0; s0 < n_part; s0 += 8) {
        float4 a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (s0 + k >= n_part) continue;                 // wave-uniform: splits that do not exist cost no request
            a[k] = *reinterpret_cast<const float4 *>(part + (size_t)(s0 + k) * part_stride + off);
            b[k] = *reinterpret_cast<const float4 *>(part + (size_t)(s0 + k) * part_stride + off + 4);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (s0 + k >= n_part) continue;
            acc[0] += a[k].x; acc[1] += a[k].y; acc[2] += a[k].z; acc[3] += a[k].w;
            acc[4] += b[k].x; acc[5] += b[k].y; acc[6] += b[k].z; acc[7] += b[k].w;
        }
    }
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; j++) o.v[j] = (T)acc[j];
    return o;
}

__device__ __forceinline__ float block_sum(float v, float *smem) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < nw; k++) t += smem[k];
    __syncthreads();
    return t;
}

// out[r] = table[tokens[r]]
template <typename T>
__global__ __launch_bounds__(256) void k_embed_rows(const int *__restrict__ tokens, const T *__restrict__ table, T *__restrict__ out,
                                                    int hidden, int vocab) {
    const int r = blockIdx.x;
    int t = tokens[r]; t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    const T *src = table + (size_t)t * hidden;
    T *dst = out + (size_t)r * hidden;
    for (int c = threadIdx.x * 8; c < hidden; c += blockDim.x * 8) st8(dst + c, ld8(src + c));
}

// the norm-fold forward's embedding: the rows, plus each row's sum of squares per 16-column tile ([hidden / 16][16] fp32, tile-major) --
// the form in which the projections that apply the RMSNorm themselves expect it (gemm_kernels.hip: NormArgs)
template <typename T>
__global__ __launch_bounds__(256) void k_embed_rows_ssq(const int *__restrict__ tokens, const T *__restrict__ table, T *__restrict__ out,
                                                        float *__restrict__ ssq, int hidden, int vocab) {
    const int r = blockIdx.x;
    int t = tokens[r]; t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    const T *src = table + (size_t)t * hidden;
    T *dst = out + (size_t)r * hidden;
    for (int tile = threadIdx.x; tile < hidden / 16; tile += blockDim.x) {
        const Vec8<T> a = ld8(src + 16 * tile), b = ld8(src + 16 * tile + 8);
        st8(dst + 16 * tile, a); st8(dst + 16 * tile + 8, b);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { const float fa = (float)a.v[j], fb = (float)b.v[j]; q += fa * fa; q += fb * fb; }
        ssq[(size_t)tile * 16 + r] = q;
    }
}

// k_embed_rows_ssq and k_rope_rows in ONE launch: the two open every norm-fold forward, neither depends on the other, and each alone is a
// launch at the latency floor (~5 us).  Blocks [0, rows): the embedding rows; the blocks behind them: cos | sin of the rope rows.
template <typename T>
__global__ __launch_bounds__(256) void k_embed_rows_ssq_rope(const int *__restrict__ tokens, const T *__restrict__ table, T *__restrict__ out,
                                                             float *__restrict__ ssq, int rows, int hidden, int vocab, const int *__restrict__ rel_pos,
                                                             const int *__restrict__ d_base, const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                             float *__restrict__ cs, int rope_rows, int max_pos) {
    if ((int)blockIdx.x >= rows) {
        const int i = ((int)blockIdx.x - rows) * blockDim.x + threadIdx.x, r = i >> 6, j = i & 63;
        if (r >= rope_rows) return;
        int pos = d_base[0] + rel_pos[r];
        pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
        cs[r * 128 + j] = cos_t[(size_t)pos * 64 + j];
        cs[r * 128 + 64 + j] = sin_t[(size_t)pos * 64 + j];
        return;
    }
    const int r = blockIdx.x;
    int t = tokens[r]; t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    const T *src = table + (size_t)t * hidden;
    T *dst = out + (size_t)r * hidden;
    for (int tile = threadIdx.x; tile < hidden / 16; tile += blockDim.x) {
        const Vec8<T> a = ld8(src + 16 * tile), b = ld8(src + 16 * tile + 8);
        st8(dst + 16 * tile, a); st8(dst + 16 * tile + 8, b);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) { const float fa = (float)a.v[j], fb = (float)b.v[j]; q += fa * fa; q += fb * fb; }
        ssq[(size_t)tile * 16 + r] = q;
    }
}


// HF LlamaRMSNorm: out = w * (x * rsqrt(mean(x^2) + eps)).to(dtype); optional fused residual add:
// x <- x + delta first (stored back), as LlamaDecoderLayer does between its two halves.
template <typename T, bool ADD>
__global__ __launch_bounds__(1024) void k_rmsnorm(T *__restrict__ x, const T *__restrict__ delta, const T *__restrict__ w,
                                                 T *__restrict__ out, int hidden, float eps, int n_part, long long part_stride, int rows, WarmArgs warm) {
    __shared__ float red[16];
    if ((int)blockIdx.x >= rows) {           // warm workgroups: the head of the next projection's weight stream -> this XCD's L2 (warm_device.h)
        const unsigned a = warm_next_projection(warm, (int)blockIdx.x - rows);
        if (a == 0x9E3779B9u && hidden < 0) out[0] = (T)0.f;      // never true: keeps the loads alive
        return;
    }
    constexpr int MAXV = 4;                                  // 8-element vectors kept in registers per thread: hidden <= 8192
    const size_t base = (size_t)blockIdx.x * hidden;
    const int stride = blockDim.x * 8;
    Vec8<T> xv[MAXV], wv[MAXV];
    float ss = 0.f;
    // one pass: the row stays in registers between the sum of squares and the scaling, and the weight loads are issued
    // before the block reduction so that their latency hides behind it
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        const int c = threadIdx.x * 8 + i * stride;
        if (c >= hidden) break;
        Vec8<T> a = ld8(x + base + c);
        wv[i] = ld8(w + c);
        if (ADD) {
            const Vec8<T> d = ld8_or_partials<T>(delta, reinterpret_cast<const float *>(delta), n_part, (size_t)part_stride, base + c);
#pragma unroll
            for (int j = 0; j < 8; j++) a.v[j] = (T)((float)a.v[j] + (float)d.v[j]);
            st8(x + base + c, a);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { const float f = (float)a.v[j]; ss += f * f; }
        xv[i] = a;
    }
    const float tot = block_sum(ss, red);
    const float rs = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        const int c = threadIdx.x * 8 + i * stride;
        if (c >= hidden) break;
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; j++) { const T h = (T)((float)xv[i].v[j] * rs); o.v[j] = (T)((float)wv[i].v[j] * (float)h); }
        st8(out + base + c, o);
    }
}

// RoPE (HF rotate_half convention, cos/sin tables fp32 [max_pos][D/2]) on q and k, then
//   q_out[r][h][:]            <- rotated q
//   k_cache[kvh][L + r][:]    <- rotated k        (SamdStaticCache.update, cache.py:106-109)
//   v_cache[kvh][L + r][:]    <- v
// position of row r = L + rel_pos[r] (tree depth or sequence offset; samd_model.py:127-132).
// grid = (rows, H + 2*Hkv), block = D/2 threads; rows >= *d_n are skipped.
template <typename T>
__global__ void k_rope_kv(const T *__restrict__ qkv, const int *__restrict__ rel_pos, const int *__restrict__ d_L,
                          const int *__restrict__ d_n, const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                          T *__restrict__ q_out, T *__restrict__ k_cache, T *__restrict__ v_cache, int H, int Hkv, int D,
                          long long max_len, int max_pos, int n_part, long long part_stride, int v_transposed, const float *__restrict__ cs) {
    const int r = blockIdx.x, hh = blockIdx.y, j = threadIdx.x, half = D >> 1;
    // cs != null: this row's cos | sin were prepared by k_rope_rows (once per forward, [rows][D]); they are requested HERE, in the same
    // round trip as the operands, instead of after L and the row's position are known (one dependent memory round trip less)
    float c_pre = 0.f, s_pre = 0.f;
    if (cs) { c_pre = cs[(size_t)r * D + j]; s_pre = cs[(size_t)r * D + half + j]; }
    // the operand loads do not depend on n / L: they are issued first, together with the scalars and this row's relative
    // position, so that the kernel is two memory round trips (operands + scalars, then cos/sin) instead of four
    const int rel = rel_pos[r], n = d_n[0], L = d_L[0];
    const size_t soff = ((size_t)r * (H + 2 * Hkv) + hh) * D;
    float x1, x2;                                              // elements j and j + D/2 of this head's row
    if (n_part == 0) { x1 = (float)qkv[soff + j]; x2 = (float)qkv[soff + j + half]; }
    else {
        const float *part = reinterpret_cast<const float *>(qkv);
        float a = 0.f, b = 0.f;
        for (int s0 = 0; s0 < n_part; s0 += 8) {                  // 16 loads in flight per group (see ld8_or_partials)
            float pa[8], pb[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (s0 + k >= n_part) continue;
                pa[k] = part[(size_t)(s0 + k) * part_stride + soff + j]; pb[k] = part[(size_t)(s0 + k) * part_stride + soff + j + half];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) if (s0 + k < n_part) { a += pa[k]; b += pb[k]; }
        }
        x1 = (float)(T)a; x2 = (float)(T)b;                    // rounded like the GEMM's own output
    }
    if (r >= n) return;
    if (L + r >= max_len) return;                              // never write past the cache (the host guard breaks earlier)
    if (hh >= H + Hkv) {                                       // V: plain copy (row-major cache, or the transposed one of samd_attention_block)
        if (!v_cache) return;
        if (v_transposed) {
            T *dst = v_cache + (size_t)(hh - H - Hkv) * D * max_len + L + r;
            dst[(size_t)j * max_len] = (T)x1; dst[(size_t)(j + half) * max_len] = (T)x2;
        } else {
            T *dst = v_cache + ((size_t)(hh - H - Hkv) * max_len + L + r) * D;
            dst[j] = (T)x1; dst[j + half] = (T)x2;
        }
        return;
    }
    float c, s;
    if (cs) { c = c_pre; s = s_pre; }
    else {
        int pos = L + rel; pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
        c = cos_t[(size_t)pos * half + j]; s = sin_t[(size_t)pos * half + j];
    }
    const T o1 = (T)(x1 * c - x2 * s), o2 = (T)(x2 * c + x1 * s);
    T *dst = hh < H ? q_out + ((size_t)r * H + hh) * D : k_cache + ((size_t)(hh - H) * max_len + L + r) * D;
    dst[j] = o1; dst[j + half] = o2;
}

// The same for the prompt's rows (round 5: 33 -> ~15 us per layer at 1.3-1.5 k rows, profiles/r05_prefill.md): head_dim 128, dtype operands, position
// tables; 8 elements of each half per lane (16-byte loads / stores) and 8 heads per 64-thread workgroup instead of one 2-byte element pair
// per thread.  Same expressions, same roundings as k_rope_kv.  grid = (rows, ceil((H + 2 Hkv) / 8)).
template <typename T>
__global__ __launch_bounds__(64) void k_rope_kv_wide(const T *__restrict__ qkv, const int *__restrict__ rel_pos, const int *__restrict__ d_L,
                                                     const int *__restrict__ d_n, const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                     T *__restrict__ q_out, T *__restrict__ k_cache, T *__restrict__ v_cache, int H, int Hkv,
                                                     long long max_len, int max_pos) {
    typedef T V8 __attribute__((ext_vector_type(8)));
    const int r = blockIdx.x, hh = blockIdx.y * 8 + (threadIdx.x >> 3), j = (threadIdx.x & 7) * 8;
    if (hh >= H + 2 * Hkv) return;
    const int rel = rel_pos[r], n = d_n[0], L = d_L[0];
    const size_t soff = ((size_t)r * (H + 2 * Hkv) + hh) * 128;
    const V8 a = *reinterpret_cast<const V8 *>(qkv + soff + j), b = *reinterpret_cast<const V8 *>(qkv + soff + 64 + j);
    if (r >= n || L + r >= max_len) return;
    if (hh >= H + Hkv) {
        if (!v_cache) return;
        T *dst = v_cache + ((size_t)(hh - H - Hkv) * max_len + L + r) * 128;
        *reinterpret_cast<V8 *>(dst + j) = a; *reinterpret_cast<V8 *>(dst + 64 + j) = b;
        return;
    }
    int pos = L + rel; pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const float4 c0 = *reinterpret_cast<const float4 *>(cos_t + (size_t)pos * 64 + j), c1 = *reinterpret_cast<const float4 *>(cos_t + (size_t)pos * 64 + j + 4);
    const float4 s0 = *reinterpret_cast<const float4 *>(sin_t + (size_t)pos * 64 + j), s1 = *reinterpret_cast<const float4 *>(sin_t + (size_t)pos * 64 + j + 4);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    V8 o1, o2;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x1 = (float)a[k], x2 = (float)b[k];
        o1[k] = (T)(x1 * c[k] - x2 * sn[k]); o2[k] = (T)(x2 * c[k] + x1 * sn[k]);
    }
    T *dst = hh < H ? q_out + ((size_t)r * H + hh) * 128 : k_cache + ((size_t)(hh - H) * max_len + L + r) * 128;
    *reinterpret_cast<V8 *>(dst + j) = o1; *reinterpret_cast<V8 *>(dst + 64 + j) = o2;
}

// The prompt's V rows into a TRANSPOSED cache (round 6): vt_cache[kvh][d][L + r] <- qkv[r][H + Hkv + kvh][d] for r < n.  One workgroup = one KV head x
// 64 rows: the rows are read as 16-byte pieces (a wave's load = 4 rows x 256 contiguous bytes), turned in LDS, and leave as 16-byte pieces of 8 keys
// of one column (a wave's store = 8 columns x 128 contiguous bytes) -- where k_rope_kv's transposed form writes one 2-byte element per lane and a
// strided torch copy (what the prefill did before) takes 17-40 us per layer at 0.5-1.5 k rows.  L % 8 != 0 or the prompt's last, partial piece of 8
// keys: element stores.  16-bit elements of either dtype (a copy).
__global__ __launch_bounds__(256) void k_v_rows_to_vt(const unsigned short *__restrict__ qkv, const int *__restrict__ d_L, const int *__restrict__ d_n,
                                                      unsigned short *__restrict__ vt_cache, int rows, int H, int Hkv, long long max_len) {
    constexpr int RS = 128 + 8;                                  // halfs per staged row: 272 B (16-byte aligned, rows 8 apart fall on different banks)
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * RS];
    const int tid = threadIdx.x, kvh = blockIdx.y, r0 = 64 * blockIdx.x;
    int n = d_n[0]; n = n < rows ? n : rows;
    const int L = d_L[0];
    const size_t row_elems = (size_t)(H + 2 * Hkv) * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i, r = u >> 4, sl = u & 15;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < n) v = *reinterpret_cast<const uint4 *>(qkv + (size_t)(r0 + r) * row_elems + (size_t)(H + Hkv + kvh) * 128 + 8 * sl);
        *reinterpret_cast<uint4 *>(&tile[r * RS + 8 * sl]) = v;
    }
    __syncthreads();
    unsigned short *col0 = vt_cache + (size_t)kvh * 128 * max_len;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i, d = u >> 3, kc8 = 8 * (u & 7);
        const long long key = (long long)L + r0 + kc8;              // cache position of this piece's first key
        int live = n - r0 - kc8; live = live > 8 ? 8 : live;
        if (key + live > max_len) live = (int)(max_len - key);      // never write past the cache (k_rope_kv: L + r >= max_len is skipped)
        if (live <= 0) continue;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = tile[(kc8 + j) * RS + d];
        unsigned short *dst = col0 + (size_t)d * max_len + key;
        if (live == 8 && (key & 7) == 0) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) if (j < live) dst[j] = e[j];
        }
    }
}

// out = silu(gate) * up, gate|up concatenated per row: gu[r] = [gate(I) | up(I)]
template <typename T>
__global__ __launch_bounds__(256) void k_silu_mul(const T *__restrict__ gu, T *__restrict__ out, int inter, int n_part, long long part_stride) {
    const size_t r = blockIdx.y;
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (c >= inter) return;
    const float *part = reinterpret_cast<const float *>(gu);
    const Vec8<T> g = ld8_or_partials<T>(gu, part, n_part, (size_t)part_stride, r * 2 * inter + c);
    const Vec8<T> u = ld8_or_partials<T>(gu, part, n_part, (size_t)part_stride, r * 2 * inter + inter + c);
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float gf = (float)g.v[j];
        const T s = (T)(gf / (1.f + __expf(-gf)));
        o.v[j] = (T)((float)s * (float)u.v[j]);
    }
    st8(out + r * inter + c, o);
}

extern "C" {

int samd_embed_rows(const int32_t *d_tokens, const void *d_table, void *d_out, int32_t rows, int32_t hidden, int32_t vocab,
                    int32_t dtype, void *stream) {
    if (!d_tokens || !d_table || !d_out || rows < 1 || hidden % 8 != 0) { samd_set_error("samd_embed_rows: invalid argument"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_embed_rows<_Float16>, dim3(rows), dim3(256), 0, st, d_tokens, (const _Float16 *)d_table, (_Float16 *)d_out, hidden, vocab);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_embed_rows<__bf16>, dim3(rows), dim3(256), 0, st, d_tokens, (const __bf16 *)d_table, (__bf16 *)d_out, hidden, vocab);
    else { samd_set_error("samd_embed_rows: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_embed_rows_ssq_rope(const int32_t *d_tokens, const void *d_table, void *d_out, float *d_ssq, int32_t rows, int32_t hidden, int32_t vocab,
                             int32_t dtype, const int32_t *d_rel_pos, const int32_t *d_base, const float *d_cos, const float *d_sin, float *d_cs,
                             int32_t rope_rows, int32_t head_dim, int32_t max_pos, void *stream) {
    if (!d_tokens || !d_table || !d_out || !d_ssq || !d_rel_pos || !d_base || !d_cos || !d_sin || !d_cs || rows < 1 || rows > 16 || hidden % 16 != 0 ||
        rope_rows < 1 || rope_rows > SAMD_MAX_DRAFT || head_dim != 128 || max_pos < 1) {
        samd_set_error("samd_embed_rows_ssq_rope: invalid argument (<= 16 embedding rows, hidden %% 16 == 0, <= 64 rope rows, head_dim 128)"); return SAMD_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(rows + (rope_rows * 64 + 255) / 256);
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_embed_rows_ssq_rope<_Float16>, grid, dim3(256), 0, st, d_tokens, (const _Float16 *)d_table, (_Float16 *)d_out, d_ssq, rows, hidden, vocab,
                                              d_rel_pos, d_base, d_cos, d_sin, d_cs, rope_rows, max_pos);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_embed_rows_ssq_rope<__bf16>, grid, dim3(256), 0, st, d_tokens, (const __bf16 *)d_table, (__bf16 *)d_out, d_ssq, rows, hidden, vocab,
                                                    d_rel_pos, d_base, d_cos, d_sin, d_cs, rope_rows, max_pos);
    else { samd_set_error("samd_embed_rows_ssq_rope: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_embed_rows_ssq(const int32_t *d_tokens, const void *d_table, void *d_out, float *d_ssq, int32_t rows, int32_t hidden, int32_t vocab,
                        int32_t dtype, void *stream) {
    if (!d_tokens || !d_table || !d_out || !d_ssq || rows < 1 || rows > 16 || hidden % 16 != 0) { samd_set_error("samd_embed_rows_ssq: invalid argument (<= 16 rows, hidden %% 16 == 0)"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_embed_rows_ssq<_Float16>, dim3(rows), dim3(256), 0, st, d_tokens, (const _Float16 *)d_table, (_Float16 *)d_out, d_ssq, hidden, vocab);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_embed_rows_ssq<__bf16>, dim3(rows), dim3(256), 0, st, d_tokens, (const __bf16 *)d_table, (__bf16 *)d_out, d_ssq, hidden, vocab);
    else { samd_set_error("samd_embed_rows_ssq: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_rmsnorm(void *d_x, const void *d_delta, const void *d_weight, void *d_out, int32_t rows, int32_t hidden, float eps,
                 int32_t dtype, int32_t n_partials, int64_t partial_stride, void *stream) {
    return samd_rmsnorm_warm(d_x, d_delta, d_weight, d_out, rows, hidden, eps, dtype, n_partials, partial_stride, nullptr, stream);
}

int samd_rmsnorm_warm(void *d_x, const void *d_delta, const void *d_weight, void *d_out, int32_t rows, int32_t hidden, float eps,
                      int32_t dtype, int32_t n_partials, int64_t partial_stride, const samd_warm_t *next, void *stream) {
    const WarmArgs wa = warm_args(next);
    const int grid = rows + warm_blocks(wa);
    const int n_part = n_partials; const long long pst = partial_stride;
    if (n_partials < 0 || (n_partials > 0 && !d_delta)) { samd_set_error("samd_rmsnorm: partials without a source"); return SAMD_E_INVALID; }
    if (!d_x || !d_weight || !d_out || rows < 1 || hidden % 8 != 0 || hidden > 8192) { samd_set_error("samd_rmsnorm: invalid argument (hidden must be a multiple of 8, <= 8192)"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    // one 8-element vector per thread where the row allows it: a row's operands (x, weight, up to 8 fp32 partial sums) are
    // then requested by twice as many waves at once; measured 3.42 -> 3.36 ms per forward at hidden 4096 (512 vs 256 threads)
    int rms_threads = ((hidden / 8 + 63) / 64) * 64;
    rms_threads = rms_threads < 64 ? 64 : (rms_threads > 1024 ? 1024 : rms_threads);
    if (dtype == SAMD_F16) {
        if (d_delta) hipLaunchKernelGGL((k_rmsnorm<_Float16, true>), dim3(grid), dim3(rms_threads), 0, st, (_Float16 *)d_x, (const _Float16 *)d_delta, (const _Float16 *)d_weight, (_Float16 *)d_out, hidden, eps, n_part, pst, rows, wa);
        else hipLaunchKernelGGL((k_rmsnorm<_Float16, false>), dim3(grid), dim3(rms_threads), 0, st, (_Float16 *)d_x, (const _Float16 *)nullptr, (const _Float16 *)d_weight, (_Float16 *)d_out, hidden, eps, n_part, pst, rows, wa);
    } else if (dtype == SAMD_BF16) {
        if (d_delta) hipLaunchKernelGGL((k_rmsnorm<__bf16, true>), dim3(grid), dim3(rms_threads), 0, st, (__bf16 *)d_x, (const __bf16 *)d_delta, (const __bf16 *)d_weight, (__bf16 *)d_out, hidden, eps, n_part, pst, rows, wa);
        else hipLaunchKernelGGL((k_rmsnorm<__bf16, false>), dim3(grid), dim3(rms_threads), 0, st, (__bf16 *)d_x, (const __bf16 *)nullptr, (const __bf16 *)d_weight, (__bf16 *)d_out, hidden, eps, n_part, pst, rows, wa);
    } else { samd_set_error("samd_rmsnorm: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

static int rope_kv_write(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                         const float *d_cos, const float *d_sin, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t rows,
                         int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t max_pos, int32_t dtype,
                         int32_t n_partials, int64_t partial_stride, int32_t v_transposed, const float *d_cs, void *stream) {
    if (d_cs && head_dim != 128) { samd_set_error("samd_rope_kv_write_cs: head_dim must be 128"); return SAMD_E_INVALID; }
    if (!d_qkv || !d_rel_pos || !d_cache_length || !d_n || ((!d_cos || !d_sin) && !d_cs) || !d_q_out || !d_k_cache || (!d_v_cache && !v_transposed) || rows < 1 ||
        head_dim % 2 != 0 || head_dim > 2048) { samd_set_error("samd_rope_kv_write: invalid argument"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (rows >= 128 && head_dim == 128 && n_partials == 0 && !d_cs && d_cos && d_sin && (dtype == SAMD_F16 || dtype == SAMD_BF16)) {
        if (v_transposed && d_v_cache) {          // the prompt's V^T columns: a tiled transposition of their own (round 6); q and K rows below
            if (max_len % 8 != 0) { samd_set_error("samd_rope_kv_write_vt: max_len must be a multiple of 8"); return SAMD_E_INVALID; }
            hipLaunchKernelGGL(k_v_rows_to_vt, dim3((rows + 63) / 64, n_kv_heads), dim3(256), 0, st, (const unsigned short *)d_qkv, d_cache_length, d_n,
                               (unsigned short *)d_v_cache, rows, n_heads, n_kv_heads, (long long)max_len);
            d_v_cache = nullptr;
        }
        const dim3 wgrid(rows, (n_heads + (d_v_cache ? 2 : 1) * n_kv_heads + 7) / 8);    // the prompt's rows: 16-byte lanes (no workgroups for V heads nobody writes)
        if (dtype == SAMD_F16) hipLaunchKernelGGL(k_rope_kv_wide<_Float16>, wgrid, dim3(64), 0, st, (const _Float16 *)d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, (_Float16 *)d_q_out, (_Float16 *)d_k_cache, (_Float16 *)d_v_cache, n_heads, n_kv_heads, (long long)max_len, max_pos);
        else hipLaunchKernelGGL(k_rope_kv_wide<__bf16>, wgrid, dim3(64), 0, st, (const __bf16 *)d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, (__bf16 *)d_q_out, (__bf16 *)d_k_cache, (__bf16 *)d_v_cache, n_heads, n_kv_heads, (long long)max_len, max_pos);
        LAUNCHCHK();
        return SAMD_OK;
    }
    const dim3 grid(rows, n_heads + 2 * n_kv_heads), block(head_dim / 2);
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_rope_kv<_Float16>, grid, block, 0, st, (const _Float16 *)d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, (_Float16 *)d_q_out, (_Float16 *)d_k_cache, (_Float16 *)d_v_cache, n_heads, n_kv_heads, head_dim, (long long)max_len, max_pos, n_partials, (long long)partial_stride, v_transposed, d_cs);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_rope_kv<__bf16>, grid, block, 0, st, (const __bf16 *)d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, (__bf16 *)d_q_out, (__bf16 *)d_k_cache, (__bf16 *)d_v_cache, n_heads, n_kv_heads, head_dim, (long long)max_len, max_pos, n_partials, (long long)partial_stride, v_transposed, d_cs);
    else { samd_set_error("samd_rope_kv_write: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_rope_kv_write(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                       const float *d_cos, const float *d_sin, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t rows,
                       int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t max_pos, int32_t dtype,
                       int32_t n_partials, int64_t partial_stride, void *stream) {
    return rope_kv_write(d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, d_q_out, d_k_cache, d_v_cache, rows, n_heads, n_kv_heads, head_dim, max_len,
                         max_pos, dtype, n_partials, partial_stride, 0, nullptr, stream);
}

// the same with the rows' cos | sin taken from d_cs (float [rows][128], samd_rope_rows -- once per forward) instead of the tables:
// the kernel is then one memory round trip long instead of two
int samd_rope_kv_write_cs(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                          const float *d_cs, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t rows,
                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype,
                          int32_t n_partials, int64_t partial_stride, void *stream) {
    return rope_kv_write(d_qkv, d_rel_pos, d_cache_length, d_n, nullptr, nullptr, d_q_out, d_k_cache, d_v_cache, rows, n_heads, n_kv_heads, head_dim, max_len,
                         1, dtype, n_partials, partial_stride, 0, d_cs, stream);
}

// the same with the V cache transposed ([H_kv][D][max_len], the layout samd_attention_block reads); d_vt_cache may be NULL (q and K only)
int samd_rope_kv_write_vt(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                          const float *d_cos, const float *d_sin, void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t rows,
                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t max_pos, int32_t dtype,
                          int32_t n_partials, int64_t partial_stride, void *stream) {
    return rope_kv_write(d_qkv, d_rel_pos, d_cache_length, d_n, d_cos, d_sin, d_q_out, d_k_cache, d_vt_cache, rows, n_heads, n_kv_heads, head_dim, max_len,
                         max_pos, dtype, n_partials, partial_stride, 1, nullptr, stream);
}

// samd_rope_kv_write_cs over the transposed V cache (round 6: the >= 32-row buckets of the runner whose verify attention is samd_tree_attention_vt)
int samd_rope_kv_write_cs_vt(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                             const float *d_cs, void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t rows,
                             int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype,
                             int32_t n_partials, int64_t partial_stride, void *stream) {
    if (!d_vt_cache) { samd_set_error("samd_rope_kv_write_cs_vt: null pointer"); return SAMD_E_INVALID; }
    return rope_kv_write(d_qkv, d_rel_pos, d_cache_length, d_n, nullptr, nullptr, d_q_out, d_k_cache, d_vt_cache, rows, n_heads, n_kv_heads, head_dim, max_len,
                         1, dtype, n_partials, partial_stride, 1, d_cs, stream);
}

int samd_silu_mul(const void *d_gate_up, void *d_out, int32_t rows, int32_t inter, int32_t dtype, int32_t n_partials,
                  int64_t partial_stride, void *stream) {
    if (!d_gate_up || !d_out || rows < 1 || inter % 8 != 0) { samd_set_error("samd_silu_mul: invalid argument"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((inter / 8 + 255) / 256, rows), block(256);
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_silu_mul<_Float16>, grid, block, 0, st, (const _Float16 *)d_gate_up, (_Float16 *)d_out, inter, n_partials, (long long)partial_stride);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_silu_mul<__bf16>, grid, block, 0, st, (const __bf16 *)d_gate_up, (__bf16 *)d_out, inter, n_partials, (long long)partial_stride);
    else { samd_set_error("samd_silu_mul: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

// causal attention of the prompt's rows (csrc/prefill_attn_device.h)
}  // extern "C"

template <bool VTS>
static int prefill_attention_impl(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t rows, int32_t pos0,
                                  int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, float scale, void *stream) {
    using namespace prefillattn;
    if (VTS && (max_len < 8 || max_len % 8 != 0 || max_len >= (1ll << 24))) {
        samd_set_error("samd_prefill_attention_vt: max_len must be a multiple of 8 below 2^24"); return SAMD_E_INVALID;
    }
    if (!d_q || !d_k_cache || !d_v_cache || !d_out || rows < 1 || pos0 < 0 || head_dim != 128 || n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 ||
        (int64_t)pos0 + rows > max_len || (dtype != SAMD_F16 && dtype != SAMD_BF16) || !(scale > 0.f)) {
        samd_set_error("samd_prefill_attention: invalid argument (head_dim 128, pos0 + rows <= max_len, f16/bf16, scale > 0)"); return SAMD_E_INVALID;
    }
    // one-time, per-device setup through samd_reserve_lds / samd_device_cus (atomic bookkeeping: first calls from several host threads may race)
    {
        static unsigned long long done_f16 = 0ull, done_bf16 = 0ull;
        const hipError_t e = dtype == SAMD_F16 ? samd_reserve_lds((const void *)k_prefill_attention<prefillattn::F16, 4, 2, 2, VTS>, 2 * LDS_BYTES, &done_f16)
                                               : samd_reserve_lds((const void *)k_prefill_attention<prefillattn::BF16, 4, 2, 2, VTS>, 2 * LDS_BYTES, &done_bf16);
        if (e != hipSuccess) { samd_set_error("samd_prefill_attention: %s", hipGetErrorString(e)); return SAMD_E_HIP; }
    }
    hipStream_t st = (hipStream_t)stream;
    const int n_blocks = (rows + QB - 1) / QB;
    const int cus = samd_device_cus();
    const int pair = n_blocks * n_heads > cus ? 1 : 0;                    // a heavy + a light row block per workgroup once the blocks outnumber the CUs
    const dim3 grid(pair ? (n_blocks + 1) / 2 : n_blocks, n_heads), block(512);                    // two key groups of 4 waves
    const float scale_log2 = scale * 1.4426950408889634f;
    if (dtype == SAMD_F16) hipLaunchKernelGGL((k_prefill_attention<prefillattn::F16, 4, 2, 2, VTS>), grid, block, 2 * LDS_BYTES, st, (const _Float16 *)d_q, (const _Float16 *)d_k_cache, (const _Float16 *)d_v_cache, (_Float16 *)d_out, rows, pos0, n_heads, n_kv_heads, (long long)max_len, scale_log2, pair);
    else hipLaunchKernelGGL((k_prefill_attention<prefillattn::BF16, 4, 2, 2, VTS>), grid, block, 2 * LDS_BYTES, st, (const __bf16 *)d_q, (const __bf16 *)d_k_cache, (const __bf16 *)d_v_cache, (__bf16 *)d_out, rows, pos0, n_heads, n_kv_heads, (long long)max_len, scale_log2, pair);
    LAUNCHCHK();
    return SAMD_OK;
}

extern "C" {

int samd_prefill_attention(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t rows, int32_t pos0,
                           int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, float scale, void *stream) {
    return prefill_attention_impl<false>(d_q, d_k_cache, d_v_cache, d_out, dtype, rows, pos0, n_heads, n_kv_heads, head_dim, max_len, scale, stream);
}

// the same over a transposed V cache, d_vt_cache [n_kv_heads][128][max_len] (round 6)
int samd_prefill_attention_vt(const void *d_q, const void *d_k_cache, const void *d_vt_cache, void *d_out, int32_t dtype, int32_t rows, int32_t pos0,
                              int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, float scale, void *stream) {
    return prefill_attention_impl<true>(d_q, d_k_cache, d_vt_cache, d_out, dtype, rows, pos0, n_heads, n_kv_heads, head_dim, max_len, scale, stream);
}

}  // extern "C"
