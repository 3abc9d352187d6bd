// eagle_kernels.hip -- the tree logic of EAGLE-2's draft expansion on the device (reference: Eagle2Model.topk_genrate,
// samd/tree_model/eagle2/eagle2_model.py:848-913 for the levels, :893-946 for the final re-rank; the reference walks Python lists
// and issues ~25 small torch ops per level).  Between two forwards of the draft head a level needs:
//   per row   log_softmax over the vocabulary and its top-k                       -> k_e2_rowstats (one workgroup per row)
//   per level cumulative scores, the top-k of the k*k candidates, the next level's inputs (token embeddings | parent
//             hidden states staged for the fc projection, ancestor mask rows)       -> k_e2_select   (one workgroup)
//   at the end the best `keep` candidates of all levels in index order + parents    -> k_e2_finish   (one workgroup)
// so a draft is ~6 launches per level instead of ~35, and the host stays ahead of the GPU without capturing PyTorch ops in a
// hipGraph.  k = 8 (top_k), depth <= 7, every top-k is (value descending, index ascending) -- torch.topk's order on tie-free rows.
// Every top-k decision is also recorded (rec_*), which is what the parity tests compare with the recorded reference.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "samd_common.h"
#include "topk_device.h"

#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

#define E2_MAXDEPTH 7
#define E2_CAND (E2_K + E2_K * E2_K * E2_MAXDEPTH)      // candidates of all levels
#define E2_SELECT_WGS 16                                // workgroups of k_e2_select (staging of the fc input rows)

// device state of one expansion (all arrays persistent; include/samd_hip.h samd_e2_state_t mirrors this)
struct E2State {
    float *row_lse;          // [8]
    float *top_logp;         // [8][8]  log-probabilities of the row's top-k
    int32_t *top_idx;        // [8][8]  their token ids
    float *scores;           // [2][8]  cumulative score of the current level's rows, double-buffered by level parity
    int32_t *cs_index;       // [8]     flat index (row * 8 + rank) of the rows chosen for the current level, in the previous level's candidates
    float *all_scores;       // [E2_CAND]
    int32_t *all_tokens;     // [E2_CAND]
    int32_t *parents_list;   // [1 + 8 * depth]
    unsigned long long *mask_rows;   // [64]  ancestor bits of the current level's rows (bit j = tree row j)
    int32_t *row_src;        // [8]     row of the previous forward's output each new row hangs off
    int32_t *ids;            // [8]     token of each new row
    // records of every top-k decision, in the reference's order
    float *rec_top_vals;     // [1 + depth][8][8]
    int32_t *rec_top_idx;    // [1 + depth][8][8]
    float *rec_best_vals;    // [depth][8]
    int32_t *rec_best_idx;   // [depth][8]
    float *rec_final_vals;   // [keep]
    int32_t *rec_final_idx;  // [keep]
};

// per row: log-sum-exp over the vocabulary and the top-8 logits (value desc, index asc) -> top_logp = logit - lse, top_idx.
// One workgroup of 1024 threads per row; 16-byte vector loads, four in flight per thread (a scalar loop with its data-dependent
// insertion branch waits a full memory latency per element: 433 us per call at a 128 k vocabulary against 15 us for this form).
// Every thread keeps the best 8 of its elements (sorted insertion); a wave merges its 64 lists by 8 rounds of wave arg-max (the
// winning lane shifts its list), wave 0 merges the 16 wave lists the same way.
template <typename T>
__global__ __launch_bounds__(1024) void k_e2_rowstats(const T *__restrict__ logits, long long vocab, long long stride, E2State S) {
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T *x = logits + (size_t)row * stride;
    float bv[E2_K]; int bi[E2_K];
#pragma unroll
    for (int k = 0; k < E2_K; k++) { bv[k] = -INFINITY; bi[k] = 0x7fffffff; }
    float m = -INFINITY, s = 0.f;
    auto take = [&](float v, int id) {
        // a -inf logit (masked vocabulary entry) adds nothing; without the guard exp(-inf - -inf) = NaN poisons the row's lse
        if (v > m) { s = s * __expf(m - v) + 1.f; m = v; } else if (v != -INFINITY) s += __expf(v - m);
        if (!e2_before(v, id, bv[E2_K - 1], bi[E2_K - 1])) return;
#pragma unroll
        for (int k = 0; k < E2_K; k++)
            if (e2_before(v, id, bv[k], bi[k])) { const float tv = bv[k]; const int ti = bi[k]; bv[k] = v; bi[k] = id; v = tv; id = ti; }
    };
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_ok = ((((size_t)x) & 15) == 0);
    const long long nvec = vec_ok ? vocab / VEC : 0;
    for (long long c0 = tid; c0 < nvec; c0 += 4 * 1024) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (c0 + 1024 * u < nvec) raw[u] = reinterpret_cast<const uint4 *>(x)[c0 + 1024 * u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (c0 + 1024 * u >= nvec) continue;
            const T *e = reinterpret_cast<const T *>(&raw[u]);
#pragma unroll
            for (int q = 0; q < VEC; q++) take(e2_f32(e[q]), (int)((c0 + 1024 * u) * VEC + q));
        }
    }
    for (long long i = nvec * VEC + tid; i < vocab; i += 1024) take(e2_f32(x[i]), (int)i);

    __shared__ float wm[16], ws[16], wv[16 * E2_K]; __shared__ int wi[16 * E2_K];
    // log-sum-exp: combine (m, s) across the wave, then across the block
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o), os = __shfl_xor(s, o);
        const float M = fmaxf(m, om);
        s = (m == -INFINITY ? 0.f : s * __expf(m - M)) + (om == -INFINITY ? 0.f : os * __expf(om - M));
        m = M;
    }
    if (lane == 0) { wm[wave] = m; ws[wave] = s; }
    // the wave's best 8: each round every lane offers the head of its sorted list, the winner shifts its list
    auto wave_top = [&](float (&hv)[E2_K], int (&hi)[E2_K], float &out_v, int &out_i) {
        float cv = hv[0]; int ci = hi[0], cl = lane;
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(cv, o); const int oi = __shfl_xor(ci, o), ol = __shfl_xor(cl, o);
            if (e2_before(ov, oi, cv, ci)) { cv = ov; ci = oi; cl = ol; }
        }
        if (cl == lane && ci != 0x7fffffff) {
#pragma unroll
            for (int k = 0; k < E2_K - 1; k++) { hv[k] = hv[k + 1]; hi[k] = hi[k + 1]; }
            hv[E2_K - 1] = -INFINITY; hi[E2_K - 1] = 0x7fffffff;
        }
        out_v = cv; out_i = ci;
    };
    for (int round = 0; round < E2_K; round++) {
        float v; int id;
        wave_top(bv, bi, v, id);
        if (lane == 0) { wv[wave * E2_K + round] = v; wi[wave * E2_K + round] = id; }
    }
    __syncthreads();
    if (wave == 0) {
        float M = -INFINITY, Ssum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) M = fmaxf(M, wm[k]);
#pragma unroll
        for (int k = 0; k < 16; k++) if (wm[k] != -INFINITY) Ssum += ws[k] * __expf(wm[k] - M);
        const float lse = M + logf(Ssum);
        if (lane == 0) S.row_lse[row] = lse;
        // 128 candidates (16 sorted wave lists): lane l < 16 owns wave l's list
        float hv[E2_K]; int hi[E2_K];
#pragma unroll
        for (int k = 0; k < E2_K; k++) { hv[k] = lane < 16 ? wv[lane * E2_K + k] : -INFINITY; hi[k] = lane < 16 ? wi[lane * E2_K + k] : 0x7fffffff; }
        for (int round = 0; round < E2_K; round++) {
            float v; int id;
            wave_top(hv, hi, v, id);
            if (lane == 0) {
                S.top_logp[row * E2_K + round] = id == 0x7fffffff ? -INFINITY : v - lse;
                S.top_idx[row * E2_K + round] = id == 0x7fffffff ? 0 : id;
            }
        }
    }
}

// The same result from MANY workgroups (the one-workgroup form above is VALU-bound on a single CU: with 64 lanes per wave some
// lane inserts at almost every element, so every element pays the 8-step insertion -- 75 us per call at a 128 k vocabulary whatever
// the row count).  Stage 1: workgroup (row, split) owns 4096 consecutive elements, 16 per thread in registers; log-sum-exp partial
// (m, s) and the segment's best 8 by 8 rounds of block arg-max (only the winning thread rescans its 16).  Stage 2: one workgroup per
// row merges the splits' (m, s) in split order and the <= 64 x 8 candidates the same way.  Order: (value desc, index asc) throughout.
template <typename T>
__global__ __launch_bounds__(256) void k_e2_rowstats_part(const T *__restrict__ logits, long long vocab, long long stride, int n_split,
                                                          float *__restrict__ part_ms, float *__restrict__ part_v, int *__restrict__ part_i) {
    const int row = blockIdx.x, split = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T *x = logits + (size_t)row * stride;
    const long long seg0 = (long long)split * E2_SEG;
    float v[E2_EPT]; int id[E2_EPT];
    e2_load_segment<T>(x, vocab, seg0, v, id);
    float m = -INFINITY, s = 0.f;
#pragma unroll
    for (int q = 0; q < E2_EPT; q++) m = fmaxf(m, v[q]);
#pragma unroll
    for (int q = 0; q < E2_EPT; q++) if (v[q] != -INFINITY) s += __expf(v[q] - m);
    {
        const float M = e2_wave_max(m);
        s = e2_wave_sum(m == -INFINITY ? 0.f : s * __expf(m - M));
        m = M;
    }
    __shared__ float wm[4], ws[4], sv[4 * E2_K]; __shared__ int si[4 * E2_K];
    if (lane == 0) { wm[wave] = m; ws[wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])), S = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) if (wm[k] != -INFINITY) S += ws[k] * __expf(wm[k] - M);
        part_ms[((size_t)row * n_split + split) * 2] = M; part_ms[((size_t)row * n_split + split) * 2 + 1] = S;
    }
    float res_v[E2_K]; int res_i[E2_K];
    e2_block_top8<E2_EPT>(v, id, res_v, res_i, sv, si);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < E2_K; k++) { part_v[((size_t)row * n_split + split) * E2_K + k] = res_v[k]; part_i[((size_t)row * n_split + split) * E2_K + k] = res_i[k]; }
    }
}

__global__ __launch_bounds__(256) void k_e2_rowstats_merge(int n_split, const float *__restrict__ part_ms, const float *__restrict__ part_v,
                                                           const int *__restrict__ part_i, E2State S) {
    const int row = blockIdx.x, tid = threadIdx.x;
    __shared__ float sv[4 * E2_K], s_lse; __shared__ int si[4 * E2_K];
    if (tid < 64) {                                           // log-sum-exp over the splits: one split per lane
        const float mk = tid < n_split ? part_ms[((size_t)row * n_split + tid) * 2] : -INFINITY;
        const float sk = tid < n_split ? part_ms[((size_t)row * n_split + tid) * 2 + 1] : 0.f;
        const float M = e2_wave_max(mk);
        const float acc = e2_wave_sum(mk == -INFINITY ? 0.f : sk * __expf(mk - M));
        if (tid == 0) s_lse = M + logf(acc);
    }
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
    e2_block_top8<2>(v, id, out_v, out_i, sv, si);            // (its barriers also publish s_lse)
    if (tid == 0) {
        const float lse = s_lse;
        S.row_lse[row] = lse;
#pragma unroll
        for (int k = 0; k < E2_K; k++) {
            S.top_logp[row * E2_K + k] = out_i[k] == 0x7fffffff ? -INFINITY : out_v[k] - lse;
            S.top_idx[row * E2_K + k] = out_i[k] == 0x7fffffff ? 0 : out_i[k];
        }
    }
}

// level = -1: the root (row 0 of top_* = the last accepted position): candidates 0..7, no selection among k*k.
// level >= 0: rows 0..7 of top_* = the outputs of tree level `level`; selects the 8 rows of level + 1 and stages their inputs.
// hidden [rows][H] = the previous forward's output states (row r of the level, or the single last accepted row for the root);
// fc_in [8][2H] <- [embed[ids[t]] | hidden[row_src[t]]]; mask_rows <- ancestor bits of the new rows.
template <typename T>
__global__ __launch_bounds__(256) void k_e2_select(E2State S, int level, const T *__restrict__ hidden, const T *__restrict__ embed, int H, int vocab,
                                                   T *__restrict__ fc_in, int32_t *__restrict__ relpos, int32_t *__restrict__ d_L, int32_t *__restrict__ d_Lw,
                                                   int32_t *__restrict__ d_n, int advance_L) {
    // Every workgroup repeats the (cheap) tree logic in its wavefront 0 and then stages its share of the fc input rows; only
    // workgroup 0 commits the state.  The one input the commit overwrites -- the rows' cumulative scores -- is double-buffered by
    // level parity (scores[0..7] / scores[8..15]), so the other workgroups never read what workgroup 0 is writing.
    __shared__ int s_ids[E2_K], s_src[E2_K];
    const int lane = threadIdx.x;
    const bool commit = blockIdx.x == 0;
    const float *scores_r = S.scores + E2_K * (level & 1);
    float *scores_w = S.scores + E2_K * ((level + 1) & 1);
    if (lane < 64) {                                      // wavefront 0 does the tree logic in lockstep (no block barrier inside)
        if (level < 0) {
            if (lane < E2_K) { s_ids[lane] = S.top_idx[lane]; s_src[lane] = 0; }
            if (lane < E2_K && commit) {
                const float sc = S.top_logp[lane];
                const int tok = S.top_idx[lane];
                scores_w[lane] = sc; S.cs_index[lane] = lane;
                S.all_scores[lane] = sc; S.all_tokens[lane] = tok;
                S.rec_top_vals[lane] = sc; S.rec_top_idx[lane] = tok;
                S.ids[lane] = tok; S.row_src[lane] = 0;
                S.mask_rows[lane] = 1ull << lane;
                if (lane == 0) S.parents_list[0] = 0;
            }
        } else {
            // every lane reads the old state first (registers), the commit below comes later in program order of the same wave
            const int r = lane >> 3;
            const float cu = S.top_logp[lane] + scores_r[r];
            const int my_tok = S.top_idx[lane];
            const int old_cs = commit ? S.cs_index[lane & 7] : 0;
            const unsigned long long prev_mask = commit ? S.mask_rows[lane & 7] : 0ull;
            const int base = E2_K + E2_K * E2_K * level;
            if (commit) {
                S.all_scores[base + lane] = cu; S.all_tokens[base + lane] = my_tok;
                S.rec_top_vals[(1 + level) * 64 + lane] = S.top_logp[lane]; S.rec_top_idx[(1 + level) * 64 + lane] = my_tok;
                // parents of this level's rows in the reference's numbering (eagle2_model.py:866-870)
                if (lane < E2_K) S.parents_list[1 + E2_K * level + lane] = old_cs + 1 + E2_K * E2_K * (level > 0 ? level - 1 : 0) + (level > 0 ? E2_K : 0);
            }
            // top-8 of the 64 cumulative scores (value desc, flat index asc)
            bool taken = false;
            float sel_v = 0.f; int sel_i = 0;
            for (int round = 0; round < E2_K; round++) {
                float bvv; int bii;
                e2_wave_best(taken ? -INFINITY : cu, taken ? 0x7fffffff : lane, bvv, bii);
                if (bii == lane) taken = true;
                if (lane == round) { sel_v = bvv; sel_i = bii; }
            }
            // lane t < 8 holds new row t: its source row, token (held by lane sel_i) and mask
            const int sel_c = lane < E2_K ? sel_i : 0;
            const int tok = __shfl(my_tok, sel_c);
            const unsigned long long pm = __shfl(prev_mask, sel_c >> 3);
            if (lane < E2_K) { s_ids[lane] = tok; s_src[lane] = sel_i >> 3; }
            if (lane < E2_K && commit) {
                const int src = sel_i >> 3;
                S.rec_best_vals[level * E2_K + lane] = sel_v; S.rec_best_idx[level * E2_K + lane] = sel_i;
                scores_w[lane] = sel_v; S.cs_index[lane] = sel_i; S.ids[lane] = tok; S.row_src[lane] = src;
                S.mask_rows[lane] = pm | (1ull << (E2_K * (level + 1) + lane));
            }
        }
    }
    if (relpos && lane < E2_K && commit) relpos[lane] = level + 1;      // depth of the new rows (position = accepted length + depth)
    if (commit && lane == 0) {                             // where the new level's K / V rows go: behind the earlier levels' rows
        if (d_L && d_Lw) {
            const int L = d_L[0] + advance_L;              // advance_L: the accepted tokens the extension forward just wrote (level -1)
            if (advance_L) d_L[0] = L;
            d_Lw[0] = L + E2_K * (level + 1);
        }
        if (d_n) d_n[0] = E2_K;
    }
    __syncthreads();
    // stage the fc projection's input rows: [embed[token] | parent hidden state]
    const int vec = H / 8;                                 // 16-byte units per half row
    const int total = E2_K * 2 * vec, nthr = blockDim.x * gridDim.x;
    for (int u0 = blockIdx.x * blockDim.x + threadIdx.x; u0 < total; u0 += 8 * nthr) {       // eight loads in flight per thread, then the stores
        uint4 buf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int u = u0 + k * nthr;
            if (u < total) {
                const int t = u / (2 * vec), c = u - t * 2 * vec;
                int tok = s_ids[t]; tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
                buf[k] = c < vec ? reinterpret_cast<const uint4 *>(embed + (size_t)tok * H)[c] : reinterpret_cast<const uint4 *>(hidden + (size_t)s_src[t] * H)[c - vec];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int u = u0 + k * nthr;
            if (u < total) { const int t = u / (2 * vec), c = u - t * 2 * vec; reinterpret_cast<uint4 *>(fc_in + (size_t)t * 2 * H)[c] = buf[k]; }
        }
    }
}

// The accepted tokens of one verified step enter the draft head without a host round trip (reference: the plugin's update() +
// the first forward of topk_genrate, eagle2.py:37-63, eagle2_model.py:835-846): row t < T of the fc projection's input is
// [embed[token after the t-th accepted token] | base-model hidden state of the t-th accepted token], where the tokens are the step's
// accepted tokens (acc_tokens[1..T-1]) followed by the bonus token (start_token), and the hidden state is the verify forward's row
// kv_index[t] (-1 = the padding entry of the reference's retrieve rows: the LAST draft row, SO/samd_model.py:144).  Also the causal
// chain mask / relative positions of the T rows and n = T.
template <typename T>
__global__ __launch_bounds__(1024) void k_e2_stage_extend(const T *__restrict__ hidden_rows, const int32_t *__restrict__ kv_index, const int32_t *__restrict__ acc_tokens,
                                                          const int32_t *__restrict__ start_token, int n_acc, int n_rows, const T *__restrict__ embed, int H, int vocab,
                                                          T *__restrict__ fc_in, int32_t *__restrict__ relpos, unsigned long long *__restrict__ mask_rows, int32_t *__restrict__ d_n,
                                                          int64_t *__restrict__ sample_token) {
    __shared__ int s_tok[64], s_row[64];
    const int tid = threadIdx.x;
    if (tid < n_acc) {
        int tok = tid + 1 < n_acc ? acc_tokens[tid + 1] : start_token[0];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        int row = kv_index[tid]; row = row < 0 ? n_rows - 1 : row; row = row < 0 ? 0 : (row > 63 ? 63 : row);
        s_tok[tid] = tok; s_row[tid] = row;
        relpos[tid] = tid;
        mask_rows[tid] = tid >= 63 ? ~0ull : ((1ull << (tid + 1)) - 1);
    }
    if (tid == 0) { d_n[0] = n_acc; if (sample_token) sample_token[0] = start_token[0]; }
    __syncthreads();
    const int vec = H / 8, total = n_acc * 2 * vec;
    for (int u0 = tid; u0 < total; u0 += 8 * blockDim.x) {
        uint4 buf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int u = u0 + k * blockDim.x;
            if (u < total) {
                const int t = u / (2 * vec), c = u - t * 2 * vec;
                buf[k] = c < vec ? reinterpret_cast<const uint4 *>(embed + (size_t)s_tok[t] * H)[c] : reinterpret_cast<const uint4 *>(hidden_rows + (size_t)s_row[t] * H)[c - vec];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int u = u0 + k * blockDim.x;
            if (u < total) { const int t = u / (2 * vec), c = u - t * 2 * vec; reinterpret_cast<uint4 *>(fc_in + (size_t)t * 2 * H)[c] = buf[k]; }
        }
    }
}

// the best `keep` of the 8 + 64 depth candidates, in candidate order, with their parents (eagle2_model.py:893-913) ->
// tokens [keep + 1] (root = sample token), parents [keep + 1] (-1 for the root)
__global__ __launch_bounds__(1024) void k_e2_finish(E2State S, int depth, int keep, const int64_t *__restrict__ sample_token, int32_t *__restrict__ out_tokens,
                                                   int32_t *__restrict__ out_parents) {
    __shared__ float sc[E2_CAND]; __shared__ int rank[E2_CAND]; __shared__ int kept[E2_CAND]; __shared__ int pos_of[E2_CAND];
    const int n = E2_K + E2_K * E2_K * depth;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sc[i] = S.all_scores[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int r = 0; const float v = sc[i];
        for (int j = 0; j < n; j++) r += e2_before(sc[j], j, v, i) ? 1 : 0;
        rank[i] = r;
        if (r < keep) { S.rec_final_vals[r] = v; S.rec_final_idx[r] = i; }
    }
    __syncthreads();
    // position of a kept candidate among the kept ones in candidate order (= sorted top indices)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int p = 0;
        for (int j = 0; j < i; j++) p += rank[j] < keep ? 1 : 0;
        pos_of[i] = p;
        if (rank[i] < keep) kept[p] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) { out_tokens[0] = (int32_t)sample_token[0]; out_parents[0] = -1; }
    for (int t = threadIdx.x; t < keep; t += blockDim.x) {
        const int i = kept[t];
        out_tokens[1 + t] = S.all_tokens[i];
        const int dp = S.parents_list[i / E2_K];          // 0 = hangs off the root; else 1 + candidate index of the parent
        int parent = 0;
        if (dp != 0) {
            // searchsorted(kept, dp - 1, left) + 1: pos_of[c] = number of kept candidates before c, whether c itself was kept or not
            int c = dp - 1; c = c < 0 ? 0 : (c >= n ? n - 1 : c);
            parent = pos_of[c] + 1;
            if (parent >= t + 1) parent = 0;               // a kept node whose parent was not kept (exact score ties only) hangs off the root
        }
        out_parents[1 + t] = parent;
    }
}

// x[r][:] = (sum_s part[s][r][:] + bias) rounded to T: the tail of the fc projection computed as split-K partials
template <typename T>
__global__ __launch_bounds__(256) void k_sum_partials_bias(const float *__restrict__ part, int n_part, long long part_stride, const T *__restrict__ bias,
                                                           T *__restrict__ out, int rows, int N) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * N) return;
    float acc = 0.f;
    for (int s = 0; s < n_part; s++) acc += part[(size_t)s * part_stride + i];
    if (bias) acc += (float)bias[i % N];
    out[i] = (T)acc;
}

extern "C" {

int64_t samd_e2_rowstats_workspace(int64_t vocab) {
    const int64_t n_split = (vocab + E2_SEG - 1) / E2_SEG;
    return n_split > E2_MAXSPLIT ? 0 : (int64_t)E2_K * n_split * (2 + 2 * E2_K) * 4;
}

int samd_e2_rowstats(const void *d_logits, int32_t dtype, int32_t rows, int64_t vocab, int64_t row_stride, const samd_e2_state_t *st, void *d_workspace,
                     int64_t workspace_bytes, void *stream) {
    if (!d_logits || !st || rows < 1 || rows > E2_K || vocab < E2_K || row_stride < vocab) { samd_set_error("samd_e2_rowstats: invalid argument"); return SAMD_E_INVALID; }
    static_assert(sizeof(E2State) == sizeof(samd_e2_state_t), "samd_e2_state_t layout"); E2State S; memcpy((void *)&S, (const void *)st, sizeof(S));
    hipStream_t s = (hipStream_t)stream;
    const int64_t need = samd_e2_rowstats_workspace(vocab);
    if (d_workspace && need > 0 && dtype != SAMD_F32) {
        if (workspace_bytes < need) { samd_set_error("samd_e2_rowstats: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)need); return SAMD_E_INVALID; }
        const int n_split = (int)((vocab + E2_SEG - 1) / E2_SEG);
        float *part_ms = (float *)d_workspace, *part_v = part_ms + (size_t)E2_K * n_split * 2;
        int *part_i = (int *)(part_v + (size_t)E2_K * n_split * E2_K);
        const dim3 grid(rows, n_split);
        if (dtype == SAMD_F16) hipLaunchKernelGGL(k_e2_rowstats_part<_Float16>, grid, dim3(256), 0, s, (const _Float16 *)d_logits, (long long)vocab, (long long)row_stride, n_split, part_ms, part_v, part_i);
        else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_e2_rowstats_part<__bf16>, grid, dim3(256), 0, s, (const __bf16 *)d_logits, (long long)vocab, (long long)row_stride, n_split, part_ms, part_v, part_i);
        else { samd_set_error("samd_e2_rowstats: bad dtype"); return SAMD_E_INVALID; }
        hipLaunchKernelGGL(k_e2_rowstats_merge, dim3(rows), dim3(256), 0, s, n_split, (const float *)part_ms, (const float *)part_v, (const int *)part_i, S);
        LAUNCHCHK();
        return SAMD_OK;
    }
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_e2_rowstats<_Float16>, dim3(rows), dim3(1024), 0, s, (const _Float16 *)d_logits, (long long)vocab, (long long)row_stride, S);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_e2_rowstats<__bf16>, dim3(rows), dim3(1024), 0, s, (const __bf16 *)d_logits, (long long)vocab, (long long)row_stride, S);
    else if (dtype == SAMD_F32) hipLaunchKernelGGL(k_e2_rowstats<float>, dim3(rows), dim3(1024), 0, s, (const float *)d_logits, (long long)vocab, (long long)row_stride, S);
    else { samd_set_error("samd_e2_rowstats: bad dtype"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_e2_select(const samd_e2_state_t *st, int32_t level, const void *d_hidden, const void *d_embed, int32_t hidden, int32_t vocab, void *d_fc_in,
                   int32_t *d_rel_pos, int32_t *d_L, int32_t *d_Lw, int32_t *d_n, int32_t advance_L, int32_t dtype, void *stream) {
    if (!st || level < -1 || level >= E2_MAXDEPTH || !d_hidden || !d_embed || !d_fc_in || hidden % 8 != 0 || vocab < 1 || advance_L < 0 || (advance_L && (!d_L || !d_Lw))) {
        samd_set_error("samd_e2_select: invalid argument"); return SAMD_E_INVALID;
    }
    static_assert(sizeof(E2State) == sizeof(samd_e2_state_t), "samd_e2_state_t layout"); E2State S; memcpy((void *)&S, (const void *)st, sizeof(S));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_e2_select<_Float16>, dim3(E2_SELECT_WGS), dim3(256), 0, s, S, level, (const _Float16 *)d_hidden, (const _Float16 *)d_embed, hidden, vocab, (_Float16 *)d_fc_in, d_rel_pos, d_L, d_Lw, d_n, advance_L);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_e2_select<__bf16>, dim3(E2_SELECT_WGS), dim3(256), 0, s, S, level, (const __bf16 *)d_hidden, (const __bf16 *)d_embed, hidden, vocab, (__bf16 *)d_fc_in, d_rel_pos, d_L, d_Lw, d_n, advance_L);
    else { samd_set_error("samd_e2_select: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_e2_stage_extend(const void *d_hidden_rows, const int32_t *d_kv_index, const int32_t *d_acc_tokens, const int32_t *d_start_token, int32_t n_accepted,
                         int32_t n_rows, const void *d_embed, int32_t hidden, int32_t vocab, void *d_fc_in, int32_t *d_rel_pos, uint64_t *d_mask_rows, int32_t *d_n,
                         int64_t *d_sample_token, int32_t dtype, void *stream) {
    if (!d_hidden_rows || !d_kv_index || !d_acc_tokens || !d_start_token || n_accepted < 1 || n_accepted > 64 || n_rows < 1 || n_rows > 64 || !d_embed || hidden % 8 != 0 ||
        vocab < 1 || !d_fc_in || !d_rel_pos || !d_mask_rows || !d_n) { samd_set_error("samd_e2_stage_extend: invalid argument"); return SAMD_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_e2_stage_extend<_Float16>, dim3(1), dim3(1024), 0, s, (const _Float16 *)d_hidden_rows, d_kv_index, d_acc_tokens, d_start_token, n_accepted,
                                              n_rows, (const _Float16 *)d_embed, hidden, vocab, (_Float16 *)d_fc_in, d_rel_pos, (unsigned long long *)d_mask_rows, d_n, d_sample_token);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_e2_stage_extend<__bf16>, dim3(1), dim3(1024), 0, s, (const __bf16 *)d_hidden_rows, d_kv_index, d_acc_tokens, d_start_token, n_accepted,
                                                    n_rows, (const __bf16 *)d_embed, hidden, vocab, (__bf16 *)d_fc_in, d_rel_pos, (unsigned long long *)d_mask_rows, d_n, d_sample_token);
    else { samd_set_error("samd_e2_stage_extend: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_e2_finish(const samd_e2_state_t *st, int32_t depth, int32_t keep, const int64_t *d_sample_token, int32_t *d_tokens, int32_t *d_parents, void *stream) {
    if (!st || depth < 1 || depth > E2_MAXDEPTH || keep < 1 || keep > E2_K + E2_K * E2_K * depth || !d_sample_token || !d_tokens || !d_parents) {
        samd_set_error("samd_e2_finish: invalid argument"); return SAMD_E_INVALID;
    }
    static_assert(sizeof(E2State) == sizeof(samd_e2_state_t), "samd_e2_state_t layout"); E2State S; memcpy((void *)&S, (const void *)st, sizeof(S));
    hipLaunchKernelGGL(k_e2_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, S, depth, keep, d_sample_token, d_tokens, d_parents);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_sum_partials_bias(const float *d_part, int32_t n_partials, int64_t partial_stride, const void *d_bias, void *d_out, int32_t rows, int32_t N,
                           int32_t dtype, void *stream) {
    if (!d_part || !d_out || n_partials < 1 || rows < 1 || N < 1) { samd_set_error("samd_sum_partials_bias: invalid argument"); return SAMD_E_INVALID; }
    const long long total = (long long)rows * N;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SAMD_F16) hipLaunchKernelGGL(k_sum_partials_bias<_Float16>, grid, block, 0, s, d_part, n_partials, (long long)partial_stride, (const _Float16 *)d_bias, (_Float16 *)d_out, rows, N);
    else if (dtype == SAMD_BF16) hipLaunchKernelGGL(k_sum_partials_bias<__bf16>, grid, block, 0, s, d_part, n_partials, (long long)partial_stride, (const __bf16 *)d_bias, (__bf16 *)d_out, rows, N);
    else { samd_set_error("samd_sum_partials_bias: dtype must be f16/bf16"); return SAMD_E_INVALID; }
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
