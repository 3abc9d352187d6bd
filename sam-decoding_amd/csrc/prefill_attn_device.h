// prefill_attn_device.h -- causal self-attention of the PROMPT's rows (the request start; what the reference runs through HF's SDPA inside
// LlamaAttention.forward on the prompt, SO/samd_model.py:102-106), hand-written for gfx950 (round 5): fused QK^T -> online softmax -> PV on
// v_mfma_f32_16x16x32 with 128 query rows x 64 keys per workgroup step, instead of PyTorch's fused SDPA (0.18 PFLOP/s on these shapes:
// 103 us per layer at 1536 rows x 32 heads, profiles/r05_prefill.md).
//
// One workgroup = one head x 128 query rows (4 waves x 32 rows = 2 fragments of 16); it walks the 64-key tiles 0 .. diagonal.  Everything is
// computed TRANSPOSED so that the softmax statistics of a query row and every accumulator of that row live in the same lane:
//   S^T[key][q] = K Q^T     A operand = K rows from LDS (lane: key l & 15, 8 consecutive d),  B operand = Q, loaded once into registers
//   O^T[d][q]  += V^T P^T   A operand = V^T from LDS (lane: d l & 15, 8 keys),                 B operand = P^T = the lane's own S^T registers
// In both products a lane's column is its query row q = l & 15: max / sum / rescale are lane-local but for one reduction over the four lanes
// (l >> 4) that share a row.  The lane's 8 P values of a 32-key step are keys {4g..4g+3} and {16+4g..16+4g+3} (g = l >> 4) -- an MFMA does not
// care which k index sits where as long as both operands agree, so V^T is read with the same permutation and P needs no shuffle at all.
// K tile: [64 keys][128 d], 16-byte slot s of row r stored at s ^ (r & 15) (ds_read_b128 of 16 rows at one slot: 16 different bank groups).
// V tile: transposed on its way into LDS, key pairs packed per dword (Vt[d][key], row stride 72 halfs).  Both staged through registers one tile
// ahead, two LDS buffers, one bare barrier per tile (LDS-only wait: the next tile's global loads stay in flight across it).  Rows of K / V at or
// beyond `total` are masked / zeroed; query rows beyond `rows` are computed on zeros and never stored.
// Measured (scripts/probes/prefill_attn_probe.hip, profiles/r05_prefill_attn.md; 32 heads, two key groups): 20.8 / 37.7 / 58.0 us at 512 / 1024 / 1536 rows against
// 33 / 52-55 / 103-107 for PyTorch's fused SDPA (which also runs 40-80 % slower on row counts that are not multiples of 128; this kernel does not
// care).  The loop is bound by instruction issue (~630 instructions per tile and wave for 64 MFMAs), not by load latency: a second register
// stage (two tiles of prefetch distance) and an 8-wave form with one fragment per wave were both measured slower.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace prefillattn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct F16 { typedef _Float16 elem; typedef half8 vec8; typedef _Float16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };
struct BF16 { typedef __bf16 elem; typedef bf16x8 vec8; typedef __bf16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };

// max / sum over the four lanes (l, l ^ 16, l ^ 32, l ^ 48) that share a query row: gfx950's permlane swaps (VALU) instead of ds_bpermute round trips.
// permlane32_swap(x, x) -> {lower half everywhere, upper half everywhere}; permlane16_swap likewise for the 16-lane rows of each half.
// (Hand-issued: through __builtin_amdgcn_permlane{32,16}_swap with the same value as both operands, hipcc 7.2 folds the two results into one
// register -- `v_permlane32_swap v3, v2; v_add v2, v3, v3` -- and every sum comes out four times too large.  The s_nop covers the VALU-write ->
// permlane-read hazard the compiler would otherwise pad.)
__device__ __forceinline__ void swap32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float row4_max(float x) {
#ifdef PA_SHFL
    x = fmaxf(x, __shfl_xor(x, 16)); return fmaxf(x, __shfl_xor(x, 32));
#endif
    float a = x, b = x;
    swap32(a, b);                              // a = {lower half, lower half}, b = {upper half, upper half}
    float c = fmaxf(a, b), d = c;
    swap16(c, d);                              // c = rows {0, 0, 2, 2}, d = rows {1, 1, 3, 3}
    return fmaxf(c, d);
}
__device__ __forceinline__ float row4_sum(float x) {
#ifdef PA_SHFL
    x += __shfl_xor(x, 16); return x + __shfl_xor(x, 32);
#endif
    float a = x, b = x;
    swap32(a, b);
    float c = a + b, d = c;
    swap16(c, d);
    return c + d;
}

#ifndef PA_ABL
#define PA_ABL 0        // diagnostic builds of scripts/probes/prefill_attn_probe.hip only (results wrong, timing only): bit 0 no exponentials, bit 1 one V^T
#endif                  // fragment read per tile instead of 16, bit 2 one K fragment read instead of 16, bit 3 no staging after the prologue
#ifdef PA_NO_SGB
#define PA_SGB(a, b, c) ((void)0)
#else
#define PA_SGB(a, b, c) __builtin_amdgcn_sched_group_barrier(a, b, c)
#endif
constexpr int QB = 128;            // query rows per workgroup
constexpr int KT = 64;             // keys per tile
constexpr int D = 128;             // head_dim
constexpr int VT_STRIDE = KT + 8;  // halfs per V^T row: 144 B keeps 8-byte alignment and spreads the banks
constexpr int K_BYTES = KT * D * 2, VT_BYTES = D * VT_STRIDE * 2;
constexpr int LDS_BYTES = 2 * (K_BYTES + VT_BYTES);       // per key group

// grid = (pair ? ceil(n_blocks / 2) : n_blocks, n_heads) with n_blocks = ceil(rows / 128): with `pair` a workgroup takes a heavy and a light
// row block (see the loop) -- worth it once there are more row blocks x heads than CUs (> 1024 rows at 32 heads); below that, more workgroups win.  Query row i sits at position pos0 + i and attends keys
// 0 .. pos0 + i; keys live in k_cache / v_cache [n_kv_heads][max_len][128]; `total` = pos0 + rows keys exist.
// NG = 2 (round 5, last): TWO groups of NW waves work on the same 128 query rows, group i on the key tiles i, i + 2, ... with its own LDS
// buffers and accumulators; at the end group 1 hands (m, l, O) to group 0 through LDS.  The chain of a row block -- what bounds the launch --
// is half as long, for one in-workgroup merge (no global partials, no second launch).  8 waves, 136 KiB of LDS: one workgroup per CU.
// NW waves x NF query fragments of 16 rows each = 128 query rows per workgroup: (4, 2) = fewer LDS reads per MFMA, (8, 1) = half the work per
// wave and tile, i.e. a shorter chain for the long row blocks (which bound the launch) and four waves per SIMD to hide it behind
// VTS (round 6): the V half of the cache is TRANSPOSED ([n_kv_heads][128][max_len], the layout samd_tree_attention_vt reads).  The LDS image of a
// V tile is the same (Vt[d][key], same XOR of the key-pair column); it is then COPIED there -- one 16-byte load and one 16-byte LDS store per
// 8 keys of a column -- where the row-major cache needs eight v_perm + eight 4-byte stores per pair of 16-byte loads.
template <typename TT, int NW, int NF, int NG = 1, bool VTS = false>
#ifndef PA_MINW8
#define PA_MINW8 4
#endif
#ifndef PA_MINW4
#define PA_MINW4 2
#endif
__global__ __launch_bounds__(64 * NW * NG, NG == 2 ? 2 : (NW == 4 ? PA_MINW4 : PA_MINW8)) void k_prefill_attention(const typename TT::elem *__restrict__ q, const typename TT::elem *__restrict__ kc,
                                                              const typename TT::elem *__restrict__ vc, typename TT::elem *__restrict__ out,
                                                              int rows, int pos0, int n_heads, int n_kv_heads, long long max_len, float scale_log2, int pair) {
    typedef typename TT::elem E;
    typedef typename TT::vec8 V8;
    typedef typename TT::vec4 V4;
    extern __shared__ __attribute__((aligned(16))) char pa_lds[];
    constexpr int NT = 64 * NW, KC = 1024 / NT, VC = 512 / NT;          // threads of a group; K units and V items per thread
    const int gi = NG == 1 ? 0 : (int)threadIdx.x / NT;                  // key group of this wave
    const int tid = (int)threadIdx.x - gi * NT, w = tid >> 6, l = tid & 63, lr = l & 15, g = l >> 4;
    char *grp_lds = pa_lds + gi * 2 * (K_BYTES + VT_BYTES);
    const int h = blockIdx.y;
    const int kvh = h / (n_heads / n_kv_heads);
    const E *kbase = kc + (size_t)kvh * max_len * D, *vbase = vc + (size_t)kvh * max_len * D;
    const int total = pos0 + rows;
    static_assert(NW * NF * 16 == QB, "128 query rows per workgroup");

    // staging registers: K tile = 1024 16-byte units, KC per thread (unit u = tid + NT c: key row u >> 4, slot u & 15); V tile: a thread's item =
    // key pair p x one 8-wide d chunk, VC items.
    // (named scalars, unconditional loads: as arrays, or requested under a condition, the compiler keeps the staging registers in scratch memory)
    uint4 st_k0, st_k1, st_k2, st_k3, st_va0, st_va1, st_vb0, st_vb1;
    // a tile's source addresses = wave-uniform tile base (scalar registers) + a per-lane byte offset fixed for the whole launch: no vector
    // arithmetic per request (the cache's last, partial tile excepted: load_stage).
    uint32_t k_off[KC], v_off[VC];
#pragma unroll
    for (int c = 0; c < KC; c++) { const int u = tid + NT * c; k_off[c] = (uint32_t)((u >> 4) * 256 + 16 * (u & 15)); }
#pragma unroll
    for (int it = 0; it < VC; it++) { const int item = tid + NT * it; v_off[it] = (uint32_t)((2 * ((item >> 3) & 31)) * 256 + 16 * ((item & 7) + 8 * (item >> 8))); }
    // VTS: piece u = tid + NT i (i < KC) = column d = u >> 3, keys 8 (u & 7) .. + 8 of the tile; byte offset d max_len 2 + 16 (u & 7) (the host checks max_len < 2^24)
    // NT is a multiple of 8: every piece of a thread has the same 8-key chunk tid & 7, and piece c lies (NT / 8) c columns below piece 0 -- ONE per-lane
    // offset; the rest of a piece's address is wave-uniform
    const uint32_t vt_off0 = VTS ? (uint32_t)(tid >> 3) * (uint32_t)max_len * 2u + 16u * (uint32_t)(tid & 7) : 0u;
    const long long vt_step = (long long)(NT >> 3) * max_len * 2;           // bytes between two pieces of a thread
    auto load_stage = [&](int key0) {
        // A tile that lies inside the cache (every tile but the last one of a cache whose length is not a multiple of 64, and the never-used
        // request past the last tile) takes the precomputed offsets.  The cache's LAST, partial tile clamps every ROW to max_len - 1 instead
        // (round 6, ADVICE r05: clamping the tile BASE shifted the keys of a needed tile while store_stage / the causal mask still indexed them
        // as key0 .., and read past a cache shorter than one tile): rows at or beyond `total` are masked (K) / zeroed (V) whatever they hold.
        long long kc0 = key0 < max_len ? key0 : max_len - 1; kc0 = kc0 < 0 ? 0 : kc0;
        const char *kt = reinterpret_cast<const char *>(kbase) + kc0 * 256, *vtb = reinterpret_cast<const char *>(vbase) + kc0 * 256;
        uint32_t ko[KC], va[VC], vb[VC];
        if (kc0 + KT <= max_len) {                                  // wave-uniform
#pragma unroll
            for (int c = 0; c < KC; c++) ko[c] = k_off[c];
#pragma unroll
            for (int it = 0; it < VC; it++) { va[it] = v_off[it]; vb[it] = v_off[it] + 256; }
        } else {
            const uint32_t lim = (uint32_t)(max_len - 1 - kc0);     // last row of the tile that exists in the cache
#pragma unroll
            for (int c = 0; c < KC; c++) { const uint32_t r = k_off[c] >> 8; ko[c] = (r < lim ? r : lim) * 256 + (k_off[c] & 255); }
#pragma unroll
            for (int it = 0; it < VC; it++) {
                const uint32_t r = v_off[it] >> 8, lo = v_off[it] & 255;
                va[it] = (r < lim ? r : lim) * 256 + lo; vb[it] = (r + 1 < lim ? r + 1 : lim) * 256 + lo;
            }
        }
        st_k0 = *reinterpret_cast<const uint4 *>(kt + ko[0]); st_k1 = *reinterpret_cast<const uint4 *>(kt + ko[1]);
        if (KC == 4) { st_k2 = *reinterpret_cast<const uint4 *>(kt + ko[2]); st_k3 = *reinterpret_cast<const uint4 *>(kt + ko[3]); }
        if constexpr (VTS) {
            const char *vcol = reinterpret_cast<const char *>(vbase) + kc0 * 2;      // wave-uniform
            long long vo = vt_off0;
            if (kc0 + KT > max_len) {                               // the cache's last, partial tile: 8-key pieces clamped into it (max_len % 8 == 0)
                const long long ch = tid & 7;
                const long long k0 = kc0 + 8 * ch + 8 <= max_len ? kc0 + 8 * ch : max_len - 8;
                vo += 2 * (k0 - kc0 - 8 * ch);                      // (may be negative: kc0 itself is clamped to the cache's last row)
            }
            st_va0 = *reinterpret_cast<const uint4 *>(vcol + vo); st_vb0 = *reinterpret_cast<const uint4 *>(vcol + vt_step + vo);
            if (KC == 4) { st_va1 = *reinterpret_cast<const uint4 *>(vcol + 2 * vt_step + vo); st_vb1 = *reinterpret_cast<const uint4 *>(vcol + 3 * vt_step + vo); }
            return;
        }
        st_va0 = *reinterpret_cast<const uint4 *>(vtb + va[0]); st_vb0 = *reinterpret_cast<const uint4 *>(vtb + vb[0]);
        if (VC == 2) { st_va1 = *reinterpret_cast<const uint4 *>(vtb + va[VC - 1]); st_vb1 = *reinterpret_cast<const uint4 *>(vtb + vb[VC - 1]); }
    };
    auto store_stage = [&](int key0, int buf) {
        char *kb_ = grp_lds + buf * (K_BYTES + VT_BYTES);
        E *vt = reinterpret_cast<E *>(kb_ + K_BYTES);
        auto k_dst = [&](int c) -> uint4 * {
            const int u = tid + NT * c, r = u >> 4, sl = u & 15;
            return reinterpret_cast<uint4 *>(kb_ + r * 256 + ((sl ^ (r & 15)) * 16));
        };
        *k_dst(0) = st_k0; *k_dst(1) = st_k1;
        if (KC == 4) { *k_dst(2) = st_k2; *k_dst(3) = st_k3; }
        auto v_put = [&](int it, uint4 ra, uint4 rb) {
            const int item = tid + NT * it, p = (item >> 3) & 31, d0 = 8 * ((item & 7) + 8 * (item >> 8));
            // the key-pair column is XORed with 4 x (d chunk & 7): a wave's 64 dword stores (8 pairs x 8 chunks, rows 144 B apart = the same
            // bank for every chunk) then fall on 32 different banks instead of 8 -- 2-way instead of 8-way conflicts; readers undo it
            const int vx = (item & 7) << 2;
            const int ka = key0 + 2 * p, kb = ka + 1;
            // rows at or beyond `total` are zero: their P is 0, but 0 x garbage could be NaN
            if (ka >= total) ra = make_uint4(0, 0, 0, 0);
            if (kb >= total) rb = make_uint4(0, 0, 0, 0);
            const unsigned int wa[4] = {ra.x, ra.y, ra.z, ra.w}, wb[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // {key 2p's element j, key 2p + 1's element j} in one v_perm_b32 (bytes 0-3 = wa, 4-7 = wb)
                *reinterpret_cast<unsigned int *>(&vt[(d0 + j) * VT_STRIDE + 2 * (p ^ vx)]) = __builtin_amdgcn_perm(wb[j >> 1], wa[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
            }
        };
        if constexpr (VTS) {
            const int ch = tid & 7;                                         // the same 8-key chunk for every piece of this thread
            if (key0 + KT > total) {                                        // wave-uniform: the prompt's last tile -- keys that do not exist are zero
                const int live = total - key0 - 8 * ch;                     // (P = 0 there, but 0 x garbage could be NaN)
                unsigned m[4];
#pragma unroll
                for (int d2 = 0; d2 < 4; d2++) m[d2] = (2 * d2 < live ? 0xFFFFu : 0u) | (2 * d2 + 1 < live ? 0xFFFF0000u : 0u);
                st_va0.x &= m[0]; st_va0.y &= m[1]; st_va0.z &= m[2]; st_va0.w &= m[3];
                st_vb0.x &= m[0]; st_vb0.y &= m[1]; st_vb0.z &= m[2]; st_vb0.w &= m[3];
                if (KC == 4) {
                    st_va1.x &= m[0]; st_va1.y &= m[1]; st_va1.z &= m[2]; st_va1.w &= m[3];
                    st_vb1.x &= m[0]; st_vb1.y &= m[1]; st_vb1.z &= m[2]; st_vb1.w &= m[3];
                }
            }
            // piece c = column d = (tid >> 3) + (NT / 8) c, key pairs 4 ch .. 4 ch + 3; the pair column XORed with 4 x (d chunk & 7) as v_put does (a multiple
            // of 4: the four pairs stay together).  (d >> 3) & 7 = (wave + (NT / 64) c) & 7: with NT = 256 it alternates between two values
            char *vrow = reinterpret_cast<char *>(vt) + (tid >> 3) * (VT_STRIDE * 2);
            auto slot = [&](int c) { return 16 * (ch ^ (((tid >> 6) + (NT >> 6) * c) & 7)) + c * (NT >> 3) * (VT_STRIDE * 2); };
            *reinterpret_cast<uint4 *>(vrow + slot(0)) = st_va0; *reinterpret_cast<uint4 *>(vrow + slot(1)) = st_vb0;
            if (KC == 4) { *reinterpret_cast<uint4 *>(vrow + slot(KC - 2)) = st_va1; *reinterpret_cast<uint4 *>(vrow + slot(KC - 1)) = st_vb1; }
            return;
        }
        v_put(0, st_va0, st_vb0);
        if (VC == 2) v_put(1, st_va1, st_vb1);
    };

    // Causal work grows with the row block: block b walks 2 (b + 1) key tiles.  A workgroup therefore takes TWO blocks, n_blocks - 1 - x and then x
    // (x = blockIdx.x < n_blocks / 2; the middle block of an odd count alone): every workgroup walks the same n_blocks + 1 tiles, where one block
    // per workgroup left the launch waiting for the last block's 2 n_blocks tiles (70 -> 65 us per layer at 1536 rows, 113 -> 86 at 2048;
    // slower below ~1100 rows, where it halves a workgroup count that is already under the CU count: the host decides).
    const int n_blocks = (rows + QB - 1) / QB;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
    const int qb = pass == 0 ? n_blocks - 1 - (int)blockIdx.x : (int)blockIdx.x;
    if (pass == 1 && (!pair || qb >= n_blocks - 1 - (int)blockIdx.x)) break;
    const int q0 = qb * QB + 16 * NF * w;                                   // this wave's first query row
    int last_key = pos0 + qb * QB + QB - 1; last_key = last_key < total - 1 ? last_key : total - 1;
    const int ntiles = last_key / KT + 1;
    // Q as the B operand: lane (q = lr, g) holds d = 8 g + 32 kk .. + 7
    V8 qf[NF][4];
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const int row = q0 + 16 * f + lr;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row < rows) raw = *reinterpret_cast<const uint4 *>(q + ((size_t)row * n_heads + h) * D + 8 * g + 32 * kk);
            qf[f][kk] = __builtin_bit_cast(V8, raw);
        }
    }
    float m_run[NF], l_run[NF];
    floatx4 o[NF][8];
#pragma unroll
    for (int f = 0; f < NF; f++) { m_run[f] = -INFINITY; l_run[f] = 0.f; }
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int df = 0; df < 8; df++) o[f][df] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // tile 0 into LDS, tile 1 into the staging registers.  Iteration t: the registers (tile t + 1, requested a whole iteration ago) go to the other
    // LDS buffer, tile t + 2 is requested into them, tile t is computed, barrier.  (A second register set = two tiles of distance was measured
    // and is slower: the loop is bound by instruction issue -- ~630 instructions per tile and wave for 64 MFMAs -- not by load latency.)
    load_stage(gi * KT);
    store_stage(gi * KT, 0);
    load_stage((gi + NG) * KT);
    __syncthreads();
    const int first_q_pos = pos0 + qb * QB;
    for (int j = 0; j * NG < ntiles; j++) {
        const int t = NG * j + gi, buf = j & 1, key0 = t * KT;          // this group's tile of the iteration (NG == 2: the last one may not exist)
        if (t + NG < ntiles && !(PA_ABL & 8)) store_stage(key0 + NG * KT, buf ^ 1);      // every wave is past the barrier that ended its reads of that buffer
        if (!(PA_ABL & 8)) load_stage(key0 + 2 * NG * KT);
        if (t < ntiles) {
        const char *kb_ = grp_lds + buf * (K_BYTES + VT_BYTES);
        const E *vt = reinterpret_cast<const E *>(kb_ + K_BYTES);
        // ---- S^T = K Q^T: 4 key fragments x 2 query fragments
        floatx4 s[4][NF];
        {
            // 16 K fragments (key fragment kf = i >> 2, d step kk = i & 3), read four ahead of the two MFMAs that consume each
            auto rd = [&](int i) -> V8 {
                const int kf = i >> 2, kk = i & 3;
                return *reinterpret_cast<const V8 *>(kb_ + (16 * kf + lr) * 256 + (((g + 4 * kk) ^ lr) * 16));
            };
            V8 ring[4] = {rd(0), rd(1), rd(2), rd(3)};
            PA_SGB(0x100, 4, 0);        // the four reads of the prologue first: the loop's groups then run four ahead
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int kf = i >> 2, kk = i & 3;
                const V8 a = ring[i & 3];
                if (i + 4 < 16 && !(PA_ABL & 4)) ring[i & 3] = rd(i + 4);
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    if (kk == 0) s[kf][f] = (floatx4){0.f, 0.f, 0.f, 0.f};
                    s[kf][f] = TT::mfma(a, qf[f][kk], s[kf][f]);
                }
                PA_SGB(0x008, NF, 0);   // the MFMAs of this fragment ...
                PA_SGB(0x100, 1, 0);    // ... then the LDS read four fragments ahead
            }
        }
        // ---- causal mask (only the tiles that reach past the block's first row, or past the last key)
        if (key0 + KT - 1 > first_q_pos || key0 + KT > total) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const int qpos = pos0 + q0 + 16 * f + lr;
#pragma unroll
                for (int kf = 0; kf < 4; kf++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int key = key0 + 16 * kf + 4 * g + r;
                        if (key > qpos || key >= total) s[kf][f][r] = -INFINITY;
                    }
            }
        }
        // ---- online softmax per query row (this lane's column), exp2 domain
        V8 pb[NF][2];                                             // P^T as the B operand: [query fragment][32-key step]
#pragma unroll
        for (int f = 0; f < NF; f++) {
            float mx = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; kf++)
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[kf][f][r]);
            mx = row4_max(mx);
            const float m_new = fmaxf(m_run[f], mx * scale_log2);
            // (a row that has seen nothing but masked keys so far -- the second key group's first tiles -- keeps m = -inf: exponentials against 0)
            const float m_use = (NG > 1 && m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_use);
            m_run[f] = m_new;
            // two scores per instruction where the ISA has a packed fp32 form (v_pk_fma_f32, v_pk_add_f32): the loop is bound by instruction issue
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v sum2 = {0.f, 0.f};
            const float2v sc2 = {scale_log2, scale_log2}, nm2 = {-m_use, -m_use};
            E pv[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; kf++)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float2v x = {s[kf][f][r], s[kf][f][r + 1]};
                    x = __builtin_elementwise_fma(x, sc2, nm2);
                    const float2v p = (PA_ABL & 1) ? x : (float2v){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    sum2 += p;
                    pv[kf][r] = (E)p[0]; pv[kf][r + 1] = (E)p[1];
                }
            const float sum = sum2[0] + sum2[1];
            l_run[f] = l_run[f] * alpha + sum;
            if (__any(alpha != 1.f)) {                           // wave-uniform: once the running maxima have settled the 32 multiplies go
#pragma unroll
                for (int df = 0; df < 8; df++) o[f][df] *= alpha;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                V8 b;
#pragma unroll
                for (int r = 0; r < 4; r++) { b[r] = pv[2 * ks][r]; b[4 + r] = pv[2 * ks + 1][r]; }
                pb[f][ks] = b;
            }
        }
        // ---- O^T += V^T P^T: 8 d fragments x 2 key steps; the lane's 8 keys of a step = {32 ks + 4 g ..+3} and {32 ks + 16 + 4 g ..+3}
        {
            auto rd = [&](int i) -> V8 {
                const int df = i >> 1, ks = i & 1;
                const int vx = ((2 * df + (lr >> 3)) & 7) << 2;                     // the writer's column swizzle (in key pairs)
                const E *row = vt + (16 * df + lr) * VT_STRIDE;
                const V4 lo = *reinterpret_cast<const V4 *>(row + 2 * ((16 * ks + 2 * g) ^ vx)), hi = *reinterpret_cast<const V4 *>(row + 2 * ((16 * ks + 8 + 2 * g) ^ vx));
                V8 a;
#pragma unroll
                for (int r = 0; r < 4; r++) { a[r] = lo[r]; a[4 + r] = hi[r]; }
                return a;
            };
            V8 ring[4] = {rd(0), rd(1), rd(2), rd(3)};
            PA_SGB(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int df = i >> 1, ks = i & 1;
                const V8 a = ring[i & 3];
                if (i + 4 < 16 && !(PA_ABL & 2)) ring[i & 3] = rd(i + 4);
#pragma unroll
                for (int f = 0; f < NF; f++) o[f][df] = TT::mfma(a, pb[f][ks], o[f][df]);
                PA_SGB(0x008, NF, 0);
                PA_SGB(0x100, 1, 0);
            }
        }
        }
        // bare s_barrier behind an LDS-only wait: the global loads just requested for the tile after next stay in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if constexpr (NG == 2) {
        // group 1 -> group 0 through group 1's (now idle) LDS buffers: per wave [68 values][64 lanes] fp32, value index major (conflict-free)
        float *xch = reinterpret_cast<float *>(pa_lds + 2 * (K_BYTES + VT_BYTES)) + (size_t)w * (NF * 34) * 64 + l;
        if (gi == 1) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                xch[(f * 34 + 0) * 64] = m_run[f]; xch[(f * 34 + 1) * 64] = l_run[f];
#pragma unroll
                for (int df = 0; df < 8; df++)
#pragma unroll
                    for (int r = 0; r < 4; r++) xch[(f * 34 + 2 + 4 * df + r) * 64] = o[f][df][r];
            }
        }
        __syncthreads();
        if (gi == 0) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const float m1 = xch[(f * 34 + 0) * 64], l1 = xch[(f * 34 + 1) * 64];
                const float m = fmaxf(m_run[f], m1);
                const float a0 = __builtin_amdgcn_exp2f(m_run[f] - m), a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - m);
                l_run[f] = l_run[f] * a0 + l1 * a1;
#pragma unroll
                for (int df = 0; df < 8; df++)
#pragma unroll
                    for (int r = 0; r < 4; r++) o[f][df][r] = o[f][df][r] * a0 + xch[(f * 34 + 2 + 4 * df + r) * 64] * a1;
            }
        }
    }
    // ---- normalise and store: lane (q = lr, g) holds d = 16 df + 4 g + r of its row
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const float lsum = row4_sum(l_run[f]);
        const int row = q0 + 16 * f + lr;
        if (row >= rows || gi != 0) continue;
        const float inv = 1.f / lsum;
        E *dst = out + ((size_t)row * n_heads + h) * D + 4 * g;
#pragma unroll
        for (int df = 0; df < 8; df++) {
            V4 v = {(E)(o[f][df][0] * inv), (E)(o[f][df][1] * inv), (E)(o[f][df][2] * inv), (E)(o[f][df][3] * inv)};
            *reinterpret_cast<V4 *>(dst + 16 * df) = v;
        }
    }
    __syncthreads();                                                        // the next block's first tile goes into LDS buffer 0
    }
}

}  // namespace prefillattn
