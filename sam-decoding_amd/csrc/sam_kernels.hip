// sam_kernels.hip -- gfx950 kernels + C ABI of the suffix-automaton draft path.
// Reference functions replaced: see include/samd_hip.h (one citation per entry point).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#include <hipcub/hipcub.hpp>
#include "sam_device.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { samd_set_error("%s: %s", #x, hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

static inline StaticDev static_view(const samd_static_t *s) {
    StaticDev v;
    if (!s) { memset(&v, 0, sizeof(v)); return v; }
    v.nodes = s->d_nodes; v.root_next = s->d_root; v.spill = s->d_spill; v.text = s->d_text;
    v.n_states = (int32_t)s->n_states; v.vocab = (int32_t)s->vocab; v.n_text = (int32_t)s->n_text; v.kind = s->kind;
    v.chain = (const uint4 *)s->d_chain; v.chain_w = samd_chain_w(s->vocab);
    v.root16 = (const uint4 *)s->d_root16; v.bigram = (const uint4 *)s->d_d1hash; v.bigram_mask = s->n_d1hash > 0 ? (uint32_t)(s->n_d1hash - 1) : 0u;
    v.rc_bits = (const uint32_t *)s->d_rc_bits;
    v.ehash = (const uint4 *)s->d_ehash; v.edge_mask = s->n_ehash > 0 ? (uint32_t)(s->n_ehash - 1) : 0u;
    v.topk_cnt = (const int32_t *)s->d_topk_cnt;
    v.hot = (const uint4 *)s->d_hot; v.blocks = s->d_hot ? (const uint4 *)s->d_blocks : nullptr; v.eb_tok_bits = samd_eb_tok_bits(s->vocab);
    return v;
}

// ================================================================================================
// batched walk: one lane per cursor, T tokens each (time-major token matrix => coalesced token loads).
// Per visited state the lane issues ONE 16-byte load in the common case (st_transfer; node word 0 holds the suffix
// link, the length and the most frequent successor) and high-degree states resolve through a hashed spill block, so
// the node-only launch generates ~1 HBM request per visited state (profiles/r01_walk_pmc.md: TCC_EA0_RDREQ == visited states)
// and runs at the memory system's scattered-request rate (~48 G requests/s on MI355X, scripts/hbm_probe.hip).  CHAIN = with chain
// words (samd_common.h): a cursor inside a non-branching run follows up to 8 tokens from ONE 16-byte load -- 0.72 requests per
// visited state, 0.377 vs 0.419 ms per launch (profiles/r02_walk_pmc.md).  Measured and not faster: a flattened one-visit-per-
// iteration state machine with LDS-staged tokens, a lane-quad variant (4 lanes x 16 B per node), two cursors per lane with both
// first loads in flight together (r02: 0.383 ms, 6 % more requests from chain words fetched for tokens that then mismatch).
// ================================================================================================
// PATH: 0 = through the nodes only; 1 = chain words + whatever tables the handle has (chosen per transition, uniformly); 2 = a handle with
// EDGE BLOCKS: that path alone (round 6)
template <int W, int PATH>
__global__ __launch_bounds__(256, 8) void k_static_walk(StaticDev S, const int32_t *cursors, int32_t *cursors_out,
                                                     const int32_t *__restrict__ tokens, int B, int T,
                                                     int32_t *__restrict__ trace, unsigned long long *__restrict__ visited_total, int lds_words) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long visited = 0;
    // which tokens have a root child: vocab bits, in LDS when the launch reserved room for them (samd_common.h, BIGRAM TABLE)
    extern __shared__ uint32_t walk_bits[];
    const uint32_t *bits = S.rc_bits;
    constexpr bool CHAIN = PATH != 0;
    if (CHAIN && lds_words > 0) {
        for (int k = threadIdx.x; k < lds_words; k += blockDim.x) walk_bits[k] = S.rc_bits[k];
        __syncthreads();
        bits = walk_bits;
    }
    if (b < B) {
        int2 c = reinterpret_cast<const int2 *>(cursors)[b];
        int idx = c.x, len = c.y;
        int tok = tokens[b];
        ChainWord cw = chain_none();
        for (int t = 0; t < T; t++) {
            const int nxt = (t + 1 < T) ? tokens[(size_t)(t + 1) * B + b] : 0;
            if (PATH == 2) visited += st_transfer_blocks<W>(S, bits, idx, len, tok, cw);
            else if (CHAIN) visited += st_transfer_chain<W>(S, bits, idx, len, tok, cw);
            else visited += st_transfer(S, idx, len, tok);
            if (trace) reinterpret_cast<int2 *>(trace)[(size_t)t * B + b] = make_int2(CHAIN ? st_resolve(S, idx) : idx, len);
            tok = nxt;
        }
        // the result of the walk: the final (index, length) of every stream -- in place (transfer_tokens), into a separate array (lookup:
        // the reference RETURNS the pair and leaves cur_index / cur_length alone, static_sam.py:122-125), or nowhere (cursors_out null)
        if (cursors_out) reinterpret_cast<int2 *>(cursors_out)[b] = make_int2(CHAIN ? st_resolve(S, idx) : idx, len);
    }
    if (visited_total) {
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if ((threadIdx.x & 63) == 0 && visited) atomicAdd(visited_total, visited);
    }
}

// ================================================================================================
// the same walk with DECOUPLED LANES (sam_device.h, "Round 6, last"): every lane keeps its own token index; one trip of the loop = one
// dependent round for the whole wave.  The lanes' tokens sit in LDS, 16 per lane at a time ([k][lane]: a lane reads only what it wrote, and the
// bank follows the lane, so the per-lane index costs no conflict), next to the child bitmap.
// ================================================================================================
#define WALK_TCHUNK 16
template <int W>
__global__ __launch_bounds__(256, 8) void k_static_walk_async(StaticDev S, const int32_t *cursors, int32_t *cursors_out,
                                                           const int32_t *__restrict__ tokens, int B, int T,
                                                           int32_t *__restrict__ trace, unsigned long long *__restrict__ visited_total, int lds_words) {
    extern __shared__ uint32_t walk_bits[];
    const int tid = threadIdx.x, b = blockIdx.x * blockDim.x + tid;
    const uint32_t *bits = S.rc_bits;
    if (lds_words > 0) {
        for (int k = tid; k < lds_words; k += blockDim.x) walk_bits[k] = S.rc_bits[k];
        __syncthreads();
        bits = walk_bits;
    }
    int32_t *tokbuf = reinterpret_cast<int32_t *>(walk_bits + lds_words);       // [WALK_TCHUNK][256]
    const bool live = b < B;
    WalkLane L;
    L.idx = 0; L.len = 0; L.tok = 0; L.mode = WM_READY; L.a = 0; L.vtok = 0; L.ref = L.pos = L.probes = 0u; L.set_len = false;
    L.cw = chain_none();
    if (live) { const int2 c = reinterpret_cast<const int2 *>(cursors)[b]; L.idx = c.x; L.len = c.y; }
    unsigned long long visited = 0;
    for (int t0 = 0; t0 < T; t0 += WALK_TCHUNK) {
        const int tn = live ? (T - t0 < WALK_TCHUNK ? T - t0 : WALK_TCHUNK) : 0;
#pragma unroll
        for (int k = 0; k < WALK_TCHUNK; k++) if (k < tn) tokbuf[k * 256 + tid] = tokens[(size_t)(t0 + k) * B + b];
        int t = 0;
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        for (;;) {
            bool done = false;
            if (L.mode != WM_READY) done = wl_step<W>(S, bits, L, e);             // the pending load has landed
            if (done) {
                visited += (unsigned)L.vtok; L.cw.ptok = L.tok; L.mode = WM_READY;
                if (trace) reinterpret_cast<int2 *>(trace)[(size_t)(t0 + t) * B + b] = make_int2(st_resolve(S, L.idx), L.len);
                t++;
            }
            if (L.mode == WM_READY && t < tn) {                                   // a free lane starts its next token
                L.tok = tokbuf[t * 256 + tid];
                if (wl_start<W>(S, bits, L)) {
                    visited += (unsigned)L.vtok; L.cw.ptok = L.tok;
                    if (trace) reinterpret_cast<int2 *>(trace)[(size_t)(t0 + t) * B + b] = make_int2(st_resolve(S, L.idx), L.len);
                    t++;
                }
            }
            const bool pending = L.mode != WM_READY;
            if (pending) e = *wl_addr(S, L);                                       // ONE load instruction per trip for the whole wave
            if (!__any(pending || t < tn)) break;
        }
    }
    if (live && cursors_out) reinterpret_cast<int2 *>(cursors_out)[b] = make_int2(st_resolve(S, L.idx), L.len);
    if (visited_total) {
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if ((threadIdx.x & 63) == 0 && visited) atomicAdd(visited_total, visited);
    }
}

// chain words from the node image: one thread per state (samd_common.h, CHAIN WORDS)
template <int W>
__global__ __launch_bounds__(256) void k_build_chain(const SamNode *__restrict__ nodes, long long n, uint4 *__restrict__ chain) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    unsigned tokw[W];
    constexpr unsigned LOW = W == 8 ? 0x7FFFu : 0x7FFFFFFFu, HI = LOW + 1u;
    bool alive = true;
#pragma unroll
    for (int j = 0; j < W; j++) {
        tokw[j] = W == 8 ? 0xFFFFu : 0xFFFFFFFFu;
        if (alive && s + j < n) {
            const int4 w0 = reinterpret_cast<const int4 *>(nodes + s + j)[0];
            if (w0.z >= 0 && (unsigned)w0.z < LOW && (long long)w0.w == s + j + 1) {
                // flag (top bit CLEAR): state s + j has this one edge only and its suffix link is a child of the root
                const bool flagged = (w0.y & SAMD_SINGLE) && w0.x > 0 && nodes[w0.x].link == 0;
                tokw[j] = (unsigned)w0.z | (flagged ? 0u : HI);
            } else alive = false;
        } else alive = false;
    }
    uint4 o;
    if constexpr (W == 8) { o.x = tokw[0] | (tokw[1] << 16); o.y = tokw[2] | (tokw[3] << 16); o.z = tokw[4] | (tokw[5] << 16); o.w = tokw[6] | (tokw[7] << 16); }
    else { o.x = tokw[0]; o.y = tokw[1]; o.z = tokw[2]; o.w = tokw[3]; }
    chain[s] = o;
}

// ================================================================================================
// single-wavefront session kernels
// ================================================================================================
enum { OP_RESET = 1, OP_ADD = 2, OP_DYN_WALK = 4, OP_ST_WALK = 8, OP_DRAFT = 16, OP_ACCEPT = 32, OP_COMMIT = 64,
       OP_DRAFT_SEQ = 128, OP_DRAFT_TREE = 256, OP_DRAFT_FIXED = 512, OP_SET_DRAFT = 1024, OP_SET_CURSORS = 2048,
       OP_BUFFERS_ONLY = 4096, OP_ACCEPT_GIVEN = 8192 };

struct StepArgs {
    int ops;
    const int32_t *tokens;      // OP_ADD / walks / OP_SET_DRAFT
    const int32_t *parents;     // OP_SET_DRAFT
    const int32_t *d_n;         // optional device-side count
    int n;
    int commit;
    int32_t *out2;              // optional (index,length) result of a walk
    const int32_t *start_token; // OP_DRAFT
    const int32_t *node_argmax; // OP_ACCEPT
    const int32_t *given;       // OP_ACCEPT_GIVEN: {best row, accept length}
    const int32_t *given_next;  // OP_ACCEPT_GIVEN: the next start token
    int index, match, start, source, type, reverse;
    int c0, c1, c2, c3;         // OP_SET_CURSORS
    int have_static;
    int only_if_deferred;       // OP_SET_DRAFT: install only when the last lookup returned type 2
    int push_report;            // after everything else: write the report block to the session's host-coherent target (SessionDev::h_report)
};

__device__ __forceinline__ void load_draft(const SessionDev &D, StepShared &sh, int &type, int &n, int &nl, int &md) {
    const int i = lane_id();
    type = D.dmeta[D_TYPE]; n = D.dmeta[D_N]; nl = D.dmeta[D_NLEAVES]; md = D.dmeta[D_MAXDEPTH];
    for (int k = i; k < n; k += WAVE) { sh.tokens[k] = D.tokens[k]; sh.parent[k] = D.parent[k]; }
    path_fill(sh, nl < SAMD_MAX_DRAFT ? (nl < 1 ? 1 : nl) : SAMD_MAX_DRAFT);
    __syncthreads();
    for (int k = i; k < nl * md; k += WAVE) {
        const int v = D.retrieve[k];
        sh.path[k / md][k % md] = v < 0 ? PATH_PAD : (unsigned char)v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_session(SessionDev D, StaticDev S, samd_params_t P, StepArgs A) {
    __shared__ StepShared sh;
    const int lane = lane_id();

    if (A.ops & OP_RESET) {
        // DynSAM.reset (dyn_sam.py:37-43) + static cursor to root (static_sam.py:127-129)
        for (uint32_t k = lane; k <= D.hmask; k += WAVE) D.hkey[k] = SAMD_HEMPTY;
        if (lane == 0) {
            D.link[0] = -1; D.length[0] = 0; D.minend[0] = 0; D.head[0] = -1; D.tail[0] = -1;
            D.text[0] = -1;
            for (int k = 0; k < M_COUNT; k++) D.meta[k] = 0;
            D.meta[M_NSTATES] = 1; D.meta[M_NTEXT] = 1;
            for (int k = 0; k < D_COUNT; k++) D.dmeta[k] = 0;
            for (int k = 0; k < V_COUNT; k++) D.verdict[k] = 0;
            for (int k = 0; k < C_COUNT; k++) D.counters[k] = 0;
            D.start_token[0] = 0; D.cache_length[0] = 0;
        }
        wave_mem_sync();
    }
    if (A.ops & OP_SET_CURSORS) {
        if (lane == 0) { D.meta[M_CUR_IDX] = A.c0; D.meta[M_CUR_LEN] = A.c1; D.meta[M_ST_IDX] = A.c2; D.meta[M_ST_LEN] = A.c3; }
        wave_mem_sync();
    }
    int n_in = A.d_n ? A.d_n[0] : A.n;
    // phase clock of the fused step (counters C_T_*): one scalar read of the 100 MHz real-time counter per phase boundary
    unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
    auto phase_done = [&](int slot) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) D.counters[slot] += (int)(t - t_prev);
        t_prev = t;
    };

    if (A.ops & OP_ACCEPT) {
        int type, n, nl, md, a, nt;
        load_draft(D, sh, type, n, nl, md);
        do_accept(D, sh, A.node_argmax, type, n, nl, md, a, nt);
        wave_mem_sync();
        phase_done(C_T_ACCEPT);
    }
    if (A.ops & OP_ACCEPT_GIVEN) {
        int type, n, nl, md, a, nt;
        load_draft(D, sh, type, n, nl, md);
        do_accept_given(D, sh, A.given, A.given_next, type, n, nl, md, a, nt);
        wave_mem_sync();
        phase_done(C_T_ACCEPT);
    }
    if (A.ops & OP_COMMIT) {
        // DraftModel.update(accepted tokens) (draft.py:62-67)
        const int a = D.verdict[V_ACCEPT];
        for (int k = lane; k < a; k += WAVE) sh.accepted[k] = D.acc_tokens[k];
        __syncthreads();
        dyn_add_tokens(D, sh.accepted, a);
        wave_mem_sync();
        phase_done(C_T_DYN);
        if (A.have_static) {
            int is = D.meta[M_ST_IDX], ms = D.meta[M_ST_LEN];
            st_transfer_tokens(S, is, ms, sh.accepted, a);
            if (lane == 0) { D.meta[M_ST_IDX] = is; D.meta[M_ST_LEN] = ms; }
        }
        wave_mem_sync();
        phase_done(C_T_STATIC);
    }
    if (A.ops & OP_ADD) dyn_add_tokens(D, A.tokens, n_in);
    if (A.ops & OP_DYN_WALK) {
        int ci = D.meta[M_CUR_IDX], cl = D.meta[M_CUR_LEN];
        for (int i = 0; i < n_in; i++) dyn_transfer(D, ci, cl, A.tokens[i]);
        if (A.commit && lane == 0) { D.meta[M_CUR_IDX] = ci; D.meta[M_CUR_LEN] = cl; }
        if (A.out2 && lane == 0) { A.out2[0] = ci; A.out2[1] = cl; }
        wave_mem_sync();
    }
    if (A.ops & OP_ST_WALK) {
        int is = D.meta[M_ST_IDX], ms = D.meta[M_ST_LEN];
        if (A.have_static) st_transfer_tokens(S, is, ms, A.tokens, n_in);
        else if (n_in > 0) { is = 0; ms = 0; }
        if (A.commit && lane == 0) { D.meta[M_ST_IDX] = is; D.meta[M_ST_LEN] = ms; }
        if (A.out2 && lane == 0) { A.out2[0] = is; A.out2[1] = ms; }
        wave_mem_sync();
    }
    if (A.ops & OP_DRAFT) {
        do_draft(D, S, A.have_static != 0, P, sh, A.start_token[0]);
        wave_mem_sync();
        phase_done(C_T_DRAFT);
    }
    if (A.ops & OP_DRAFT_SEQ) {
        const int n = seq_draft_var(sh, D.text, D.meta[M_NTEXT], D.minend[A.index], draft_size(A.match, P.alpha, P.max_predicts), A.start);
        int nl, mxd; build_buffers(sh, n, 0, nl, mxd);
        store_draft(D, sh, 0, n, nl, mxd, A.index, A.match, 0, 0, 0);
    }
    if (A.ops & OP_DRAFT_TREE) {
        const int n = tree_draft(sh, S, A.index, draft_size(A.match, P.alpha, P.max_predicts), P.K, A.start);
        int nl, mxd; build_buffers(sh, n, 0, nl, mxd);
        store_draft(D, sh, 1, n, nl, mxd, 0, 0, A.index, A.match, 0);
    }
    if (A.ops & OP_DRAFT_FIXED) {
        int n;
        if (A.source == 0) {
            int a = A.index;
            if (a != 0) {
                const int maxlen = D.meta[M_MAXLEN];
                int to_end = maxlen - D.minend[a];
                while (D.link[a] != 0 && P.n_predicts > to_end) { a = D.link[a]; to_end = maxlen - D.minend[a]; }
            }
            n = seq_draft_fixed(sh, D.text, D.meta[M_NTEXT], D.minend[a], P.n_predicts, A.start);
            if (lane == 0) D.dmeta[9] = a;
        } else {
            n = seq_draft_fixed(sh, S.text, S.n_text, S.nodes[A.index].aux, P.n_predicts, A.start);
        }
        int nl, mxd; build_buffers(sh, n, 0, nl, mxd);
        store_draft(D, sh, 0, n, nl, mxd, A.index, 0, A.index, 0, 0);
    }
    if ((A.ops & OP_SET_DRAFT) && !(A.only_if_deferred && D.dmeta[D_TYPE] != 2)) {
        for (int k = lane; k < A.n; k += WAVE) { sh.tokens[k] = A.tokens[k]; sh.parent[k] = A.parents[k]; }
        __syncthreads();
        int nl, mxd; build_buffers(sh, A.n, A.reverse, nl, mxd);
        store_draft(D, sh, A.type, A.n, nl, mxd, 0, 0, 0, 0, A.reverse);
    }
    if (A.push_report && D.h_report) {
        // the report block as the host reads it, straight into host-coherent memory: the words first (system-scope stores), then -- behind
        // a system-scope release -- the sequence number the host polls.  One wave, so the fence orders every lane's stores.
        wave_mem_sync();
        for (int k = lane; k < SAMD_REPORT_INTS; k += WAVE)
            __hip_atomic_store(D.h_report + k, D.dmeta[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int seq = D.push_seq[0] + 1;
            D.push_seq[0] = seq;
            __hip_atomic_store(D.h_report + SAMD_REPORT_INTS, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// standalone gen_buffers on a parent array
__global__ __launch_bounds__(64) void k_tree_buffers(const int32_t *parent, int n, int reverse, int32_t *position, uint64_t *mask,
                                                     uint8_t *mask_bool, int32_t *retrieve, int32_t *shape) {
    __shared__ StepShared sh;
    const int i = lane_id();
    for (int k = i; k < n; k += WAVE) sh.parent[k] = parent[k];
    __syncthreads();
    int nl, md; build_buffers(sh, n, reverse, nl, md);
    for (int k = i; k < n; k += WAVE) {
        if (position) position[k] = sh.position[k];
        if (mask) { mask[k] = sh.mask[k]; if (n > 64) mask[n + k] = sh.mask_hi[k]; }      // n > 64: the n high words follow the n low words
        if (mask_bool) for (int j = 0; j < n; j++) mask_bool[(size_t)k * n + j] = (uint8_t)(((j < 64 ? sh.mask[k] >> j : sh.mask_hi[k] >> (j - 64))) & 1ull);
    }
    if (retrieve) for (int k = i; k < nl * md; k += WAVE) { const unsigned char v = sh.path[k / md][k % md]; retrieve[k] = v == PATH_PAD ? -1 : (int)v; }
    if (shape && i == 0) { shape[0] = nl; shape[1] = md; }
}

// scripted verifier (tests / smoke / bench): device twin of tests/scripted_lm.py next_token().
// ctx(i) = committed history + tokens on the root->i path; arg-max(i) = target[len(ctx)] while ctx is a
// prefix of target, else 3 + hash(last three ctx tokens) % (V-3).
__global__ __launch_bounds__(64) void k_scripted_argmax(SessionDev D, const int32_t *__restrict__ target, int n_target, int vocab,
                                                        int32_t *__restrict__ out) {
    __shared__ int tok[SAMD_MAX_DRAFT], par[SAMD_MAX_DRAFT];
    const int i = lane_id();
    const int n = D.dmeta[D_N];
    const int nc = D.meta[M_NTEXT] - 1;                       // committed tokens (text[0] is the sentinel)
    const int32_t *hist = D.text + 1;
    for (int k = i; k < n; k += WAVE) { tok[k] = D.tokens[k]; par[k] = D.parent[k]; }
    __syncthreads();
    // is the committed history a prefix of target?
    int bad = 0;
    for (int k = i; k < nc; k += WAVE) bad |= (k >= n_target || hist[k] != target[k]);
    const bool hist_ok = __ballot(bad != 0) == 0ull;
    for (int node = i; node < SAMD_MAX_DRAFT; node += WAVE) {         // one lane per node, two rounds above 64 nodes
        int res = 0;
        if (node < n) {
            int path[SAMD_MAX_DRAFT]; int d = 0;
            for (int j = node; j != -1; j = par[j]) path[d++] = tok[j];   // leaf -> root
            bool ok = hist_ok;
            for (int k = 0; k < d && ok; k++) { const int pos = nc + k; ok = pos < n_target && path[d - 1 - k] == target[pos]; }
            const int len = nc + d;
            if (ok && len < n_target) res = target[len];
            else {
                long long h = 1469598103ll;
                for (int k = (len >= 3 ? len - 3 : 0); k < len; k++) {
                    const int t = k < nc ? hist[k] : path[d - 1 - (k - nc)];
                    h = (h * 1000003ll + t + 7) % 2147483647ll;
                }
                res = 3 + (int)(h % (vocab - 3));
            }
        }
        out[node] = res;
    }
}

// stream-major entry points: d_tokens int32 [B][T] (what a caller that holds B token sequences has), d_trace int32 [B][T][2].  The walk
// kernel wants time-major tokens (a wavefront's 64 lanes then read 256 contiguous bytes per token index), so the matrix is transposed
// on the device first (LDS tiles, both sides coalesced: ~25 us for 2^20 x 16) and the trace back afterwards.  A kernel that reads
// stream-major rows directly -- lanes decoupled, one flattened state machine per lane, cursors handed out dynamically inside a wave --
// was built and measured in round 3 (commit af39a33, profiles/r03_walk.md): identical results, but 2.6x the instructions and 0.57 ms
// per launch against 0.33 ms for this path.
template <int WIDTH>
__global__ __launch_bounds__(256) void k_transpose_i32(const int32_t *__restrict__ src, int32_t *__restrict__ dst, int rows, int cols) {   // dst[c][r] = src[r][c]
    __shared__ int32_t tile[WIDTH][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8 threads, 32 x 32 tile
    const long long r0 = (long long)blockIdx.y * 32, c0 = (long long)blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const long long r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols)
            for (int k = 0; k < WIDTH; k++) tile[k][j][tx] = src[((size_t)r * cols + c) * WIDTH + k];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const long long c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols)
            for (int k = 0; k < WIDTH; k++) dst[((size_t)c * rows + r) * WIDTH + k] = tile[k][tx][j];
    }
}

// scripted verifier, logits form (bench.py --variant token_recycle): besides the arg-max, a model that continues the text also RANKS
// the plausible continuations -- which is what Token Recycle learns from (token_recycle.py:40-48: top-8 of every verified row).  For
// draft node i with context (a, b) = (its parent's token | the last committed token, its own token) the row gets, on top of the model's
// own logits: the scripted arg-max (samd_scripted_argmax) as the best entry, then the sparse order-2 Markov source's four successors of
// (a, b) in rank order (bench._succ, the same hash) -- the distribution the synthetic corpus and requests are drawn from.
// ORDER = 2: the four successors of (a, b) in bench._succ's order; ORDER = 1 (bench.py --variant token_recycle, round 4): the eight
// successors of b alone in bench._succ1's order over a hot vocabulary of `markov_vocab` ids -- a source whose next token depends on the
// last token, which is what a token-keyed successor table (Token Recycle) can learn.
template <typename T, int ORDER>
__global__ __launch_bounds__(64) void k_scripted_logits(SessionDev D, const int32_t *__restrict__ argmax, T *__restrict__ logits, long long stride,
                                                        int markov_vocab, int rows) {
    const int i = blockIdx.x, n = D.dmeta[D_N];
    constexpr int NS = ORDER == 2 ? 4 : 8;
    if (i >= n || i >= rows || threadIdx.x > NS) return;
    const int b = D.tokens[i], par = D.parent[i];
    const int nc = D.meta[M_NTEXT] - 1;
    const int a = par >= 0 ? D.tokens[par] : (nc > 0 ? D.text[nc] : 0);           // text[0] is the sentinel: text[nc] = last committed token
    T *row = logits + (size_t)i * stride;
    const int c = (int)threadIdx.x - 1;
    if (c < 0) { row[argmax[i]] = (T)96.f; return; }
    unsigned long long h = ((ORDER == 2 ? (unsigned long long)(unsigned)a * 1000003ull : 0ull) + (unsigned long long)(unsigned)b * 10007ull
                            + (unsigned long long)c * 7919ull + 12345ull) & 0x7FFFFFFFull;
    h = (h * 2654435761ull) & 0xFFFFFFFFull;
    const int tok = 3 + (int)(h % (unsigned long long)(markov_vocab - 3));
    if (tok != argmax[i]) row[tok] = (T)(64.f - 4.f * (float)c);
}

// gen_candidates' gather for the draft held in the session (samd_sam_only/utils.py:92-96): candidate_tokens = (tokens + [0])[retrieve]
// as int64 [leaves][depth], and the retrieve table itself as the cell -> node map of samd_posterior_sampled_nodes
__global__ __launch_bounds__(256) void k_session_candidates(SessionDev D, long long *__restrict__ cand, int32_t *__restrict__ rowmap, int cap) {
    const int nl = D.dmeta[D_NLEAVES], md = D.dmeta[D_MAXDEPTH];
    for (int k = threadIdx.x; k < nl * md && k < cap; k += blockDim.x) {
        const int v = D.retrieve[k];
        rowmap[k] = v;
        cand[k] = v < 0 ? 0 : (long long)D.tokens[v];
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" {

int samd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int samd_host_wait_spin(int32_t device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return SAMD_E_NODEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return SAMD_E_HIP;
    return hipSetDeviceFlags(hipDeviceScheduleSpin) == hipSuccess ? SAMD_OK : SAMD_E_HIP;
}

int samd_device_info(int64_t out[4]) {
    if (!out) return SAMD_E_INVALID;
    out[0] = SAMD_ABI_VERSION; out[1] = out[2] = out[3] = 0;
    int dev = 0;
    if (samd_device_count() < 1) return SAMD_E_NODEVICE;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev));
    out[1] = p.multiProcessorCount; out[2] = p.warpSize; out[3] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
    return SAMD_OK;
}

// SAMD_WALK_CHAIN=0 (read once) keeps every transition on the nodes: the A/B switch of profiles/r02_walk_pmc.md
static void launch_walk(const samd_static_t *sam, int blocks, int threads, hipStream_t st, const int32_t *d_cursors, int32_t *d_out, const int32_t *d_tokens, int B, int T,
                        int32_t *d_trace, unsigned long long *d_visited) {
    static const bool use_chain = [] { const char *e = getenv("SAMD_WALK_CHAIN"); return !(e && e[0] == '0'); }();
    // the decoupled-lane form (k_static_walk_async) is an experiment, OFF by default: -39 % load instructions, +39 % vector instructions, 0.335 vs
    // 0.301 ms on the Zipfian corpus, 0.171 vs 0.151 on the headline one (profiles/r06_walk.md section 4b).  Read per launch: tests run both forms.
    const char *env_async = getenv("SAMD_WALK_ASYNC");
    const bool use_async = env_async && env_async[0] == '1';
    const StaticDev v = static_view(sam);
    // the child bitmap rides in LDS when it is small enough to leave the occupancy alone (8 workgroups of 256 per CU: 160 KiB / 8)
    const int bit_words = v.rc_bits ? (int)((v.vocab + 31) / 32) : 0;
    const int lds_words = bit_words * 4 <= 20480 ? bit_words : 0;
    if (!use_chain || !v.chain) hipLaunchKernelGGL((k_static_walk<8, 0>), dim3(blocks), dim3(threads), 0, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, 0);
    else if (v.blocks && v.bigram && use_async && threads == 256) {
        // decoupled lanes (k_static_walk_async): the lanes' tokens ride in LDS beside the child bitmap (16 KiB + <= 4 KiB: 8 workgroups per CU);
        // a bitmap too large for that stays in memory
        const int lw = (size_t)lds_words * 4 + WALK_TCHUNK * 256 * 4 <= 20480 ? lds_words : 0;
        const size_t lds = (size_t)lw * 4 + WALK_TCHUNK * 256 * 4;
        if (v.chain_w == 8) hipLaunchKernelGGL((k_static_walk_async<8>), dim3(blocks), dim3(threads), lds, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lw);
        else hipLaunchKernelGGL((k_static_walk_async<4>), dim3(blocks), dim3(threads), lds, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lw);
    }
    else if (v.blocks && v.bigram && v.chain_w == 8) hipLaunchKernelGGL((k_static_walk<8, 2>), dim3(blocks), dim3(threads), (size_t)lds_words * 4, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lds_words);
    else if (v.blocks && v.bigram) hipLaunchKernelGGL((k_static_walk<4, 2>), dim3(blocks), dim3(threads), (size_t)lds_words * 4, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lds_words);
    else if (v.chain_w == 8) hipLaunchKernelGGL((k_static_walk<8, 1>), dim3(blocks), dim3(threads), (size_t)lds_words * 4, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lds_words);
    else hipLaunchKernelGGL((k_static_walk<4, 1>), dim3(blocks), dim3(threads), (size_t)lds_words * 4, st, v, d_cursors, d_out, d_tokens, B, T, d_trace, d_visited, lds_words);
}

// bigram table (samd_common.h): count the root children's edges, then fill.  One thread per vocabulary id.
__global__ __launch_bounds__(256) void k_bg_count(const SamNode *__restrict__ nodes, const int32_t *__restrict__ root_next, int vocab, unsigned long long *__restrict__ total) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long d = 0;
    if (tok < vocab) { const int dst = root_next[tok]; if (dst > 0) d = (unsigned long long)nodes[dst].deg; }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(total, d);
}
__global__ __launch_bounds__(256) void k_bg_fill(const SamNode *__restrict__ nodes, const SamEdge *__restrict__ spill, const int32_t *__restrict__ root_next,
                                                 int vocab, const uint4 *__restrict__ chain, uint4 *__restrict__ root16, uint4 *__restrict__ table, uint32_t mask,
                                                 uint32_t *__restrict__ rc_bits, int W, int with_hub, const uint32_t *__restrict__ bref) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= vocab) return;
    const int dst = root_next[tok];
    const unsigned length = dst > 0 ? (unsigned)(nodes[dst].length & SAMD_LEN_MASK) : 0u;
    root16[tok] = make_uint4((unsigned)dst, 0u, 0u, length);
    if (dst <= 0) return;
    atomicOr(rc_bits + (tok >> 5), 1u << (tok & 31));
    const unsigned lb = length - 1 < 3u ? length - 1 : 3u;
    // slots are claimed by compare-and-swap on the key word(s); the other words follow (a reader only runs after the launch)
    auto put = [&](int t, int d) {
        if (t < 0) return;
        uint32_t h = samd_bigram_hash(tok, t) & mask;
        uint4 c = d > 0 ? chain[d] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        unsigned hub = (with_hub && d > 0 && !(nodes[d].length & SAMD_SINGLE)) ? 0x80000000u : 0u;     // dst is branching: it is in the edge table
        if (bref) {                                          // EDGE BLOCKS (round 6): hub = dst owns a block, whose reference takes the chain word's place;
            hub = (d > 0 && bref[d]) ? 0x80000000u : 0u;     // a dst that is a root child is flagged (the cursor takes the unresolved form)
            if (hub) c.x = bref[d];
            if (d > 0 && nodes[d].link == 0) hub |= SAMD_EB_ROOTCHILD;
        }
        if (W == 8) {
            const unsigned key = (unsigned)tok | ((unsigned)t << 15) | (lb << 30);
            for (;;) {
                unsigned *kp = reinterpret_cast<unsigned *>(table + h);
                const unsigned old = atomicCAS(kp, 0xFFFFFFFFu, key);
                if (old == 0xFFFFFFFFu) { kp[1] = (unsigned)d | hub; kp[2] = c.x; kp[3] = c.y; return; }
                if (old == key) return;                                                                   // (the spill head repeats ranks 5..7)
                h = (h + 1) & mask;
            }
        } else {
            const unsigned long long key = (unsigned long long)((unsigned)tok | ((lb & 1u) << 31)) | ((unsigned long long)((unsigned)t | ((lb >> 1) << 31)) << 32);
            for (;;) {
                unsigned long long *kp = reinterpret_cast<unsigned long long *>(table + h);
                const unsigned long long old = atomicCAS(kp, ~0ull, key);
                if (old == ~0ull) { unsigned *wp = reinterpret_cast<unsigned *>(kp); wp[2] = (unsigned)d | hub; wp[3] = c.x; return; }
                if (old == key) return;
                h = (h + 1) & mask;
            }
        }
    };
    const int *w = reinterpret_cast<const int *>(nodes + dst);
    for (int k = 0; k < SAMD_INLINE_EDGES; k++) put(w[SAMD_EDGE_WORD(k)], w[SAMD_EDGE_WORD(k) + 1]);
    const int deg = w[5];
    if (deg > SAMD_INLINE_EDGES) {
        const SamEdge *sp = spill + w[14];
        const uint32_t slots = samd_spill_slots(deg);
        for (uint32_t k = 0; k < SAMD_SPILL_HEAD + slots; k++) put(sp[k].tok, sp[k].dst);
    }
}

// SAMD_BG_DISPLACED (samd_common.h): after the fill, every entry that does not sit in its home slot marks that slot.  A separate launch: the
// fill claims slots by compare-and-swap and writes the other words plainly, so the bit can only be OR-ed in once every entry is complete.
__global__ __launch_bounds__(256) void k_bg_displaced(uint4 *__restrict__ table, uint32_t mask, int W) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > mask) return;
    const uint4 e = table[p];
    if (e.x == 0xFFFFFFFFu) return;
    const int a = W == 8 ? (int)(e.x & 0x7FFFu) : (int)(e.x & 0x7FFFFFFFu), b = W == 8 ? (int)((e.x >> 15) & 0x7FFFu) : (int)(e.y & 0x7FFFFFFFu);
    const uint32_t h = samd_bigram_hash(a, b) & mask;
    if (h != p) atomicOr(reinterpret_cast<unsigned *>(table + h) + (W == 8 ? 1 : 2), SAMD_BG_DISPLACED);
}

// top-k counts (samd_common.h): one thread per (state, rank)
__global__ __launch_bounds__(256) void k_topk_counts(const SamNode *__restrict__ nodes, const SamEdge *__restrict__ spill, long long n_states, int32_t *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states * SAMD_TOPK) return;
    const long long s = i / SAMD_TOPK; const int k = (int)(i % SAMD_TOPK);
    const int *w = reinterpret_cast<const int *>(nodes + s);
    int dst = -1;
    if (k < w[5]) dst = k < SAMD_INLINE_EDGES ? w[SAMD_EDGE_WORD(k) + 1] : spill[w[14] + k - SAMD_INLINE_EDGES].dst;
    out[i] = dst >= 0 ? nodes[dst].aux : 0;
}

// edge table (samd_common.h, EDGE TABLE): count the edges of the branching states, then fill.  One thread per state.
__global__ __launch_bounds__(256) void k_eh_count(const SamNode *__restrict__ nodes, long long n, unsigned long long *__restrict__ total) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long d = 0;
    if (s >= 1 && s < n) { const int deg = nodes[s].deg; if (deg >= 2) d = (unsigned long long)deg; }
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(total, d);
}
__global__ __launch_bounds__(256) void k_eh_fill(const SamNode *__restrict__ nodes, const SamEdge *__restrict__ spill, long long n, const uint4 *__restrict__ chain,
                                                 uint4 *__restrict__ table, uint32_t mask) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < 1 || s >= n) return;
    const int *w = reinterpret_cast<const int *>(nodes + s);
    const int deg = w[5];
    if (deg < 2) return;
    auto put = [&](int t, int d) {
        if (t < 0 || d < 0) return;
        uint32_t h = samd_edge_hash((int)s, t) & mask;
        const unsigned hub = !(nodes[d].length & SAMD_SINGLE) ? 0x80000000u : 0u;
        const unsigned long long key = (unsigned long long)(unsigned)s | ((unsigned long long)(unsigned)t << 32);
        for (;;) {
            unsigned long long *kp = reinterpret_cast<unsigned long long *>(table + h);
            const unsigned long long old = atomicCAS(kp, ~0ull, key);
            if (old == ~0ull) { unsigned *wp = reinterpret_cast<unsigned *>(kp); wp[2] = (unsigned)d | hub; wp[3] = chain[d].x; return; }
            if (old == key) return;                                                                      // (the spill head repeats ranks 5..7)
            h = (h + 1) & mask;
        }
    };
    for (int k = 0; k < SAMD_INLINE_EDGES; k++) put(w[SAMD_EDGE_WORD(k)], w[SAMD_EDGE_WORD(k) + 1]);
    if (deg > SAMD_INLINE_EDGES) {
        const SamEdge *sp = spill + w[14];
        const uint32_t slots = samd_spill_slots(deg);
        for (uint32_t k = 0; k < SAMD_SPILL_HEAD + slots; k++) put(sp[k].tok, sp[k].dst);
    }
}

// ---- EDGE BLOCKS + HOT WORDS (samd_common.h, round 6) -------------------------------------------------------------------------------
// slots of state s's block (0: none -- the root, root children, states with fewer than two edges); lengths / indices that do not fit raise `bad`
__global__ __launch_bounds__(256) void k_eb_size(const SamNode *__restrict__ nodes, long long n, int per, uint32_t *__restrict__ sizes,
                                                 unsigned long long *__restrict__ total, unsigned *__restrict__ bad) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    if (s < n) {
        const SamNode &nd = nodes[s];
        if ((unsigned)(nd.length & SAMD_LEN_MASK) > SAMD_EB_IDX_MASK - 1u) atomicOr(bad, 1u);
        if (s >= 1 && nd.deg >= 2 && nd.link != 0) { m = 4; while (m < (unsigned long long)per * (unsigned)nd.deg) m <<= 1; }
        if (m > (1ull << 26)) { atomicOr(bad, 2u); m = 0; }
        sizes[s] = (uint32_t)m;
    }
    unsigned long long c = m ? 1ull : 0ull;
    for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o); c += __shfl_xor(c, o); }
    if ((threadIdx.x & 63) == 0 && m) { atomicAdd(total, m); atomicAdd(total + 1, c); }
}
// block references from the sizes and their exclusive prefix sums (in place over the offsets); rctok[root child] = its token
__global__ __launch_bounds__(256) void k_eb_ref(const uint32_t *__restrict__ sizes, uint32_t *__restrict__ off_to_ref, long long n) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t m = sizes[s];
    off_to_ref[s] = m ? ((off_to_ref[s] >> 2) | ((uint32_t)(31 - __clz(m)) << 27)) : 0u;       // first slot in units of 4 (samd_common.h)
}
__global__ __launch_bounds__(256) void k_eb_rctok(const int32_t *__restrict__ root_next, int vocab, int32_t *__restrict__ rctok) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    if (tok < vocab) { const int d = root_next[tok]; if (d > 0) rctok[d] = tok; }
}
// the fail header of state s: what its suffix link is and what a hop there needs (samd_common.h); lmax = the largest length the field holds
__device__ __forceinline__ void eb_fail(const SamNode *__restrict__ nodes, const uint32_t *__restrict__ bref, const int32_t *__restrict__ rctok, long long s,
                                        unsigned lmax, unsigned &kind, unsigned &ref, unsigned &len) {
    const int p = nodes[s].link;
    if (p <= 0) { kind = SAMD_FK_ROOT; ref = 0u; len = 0u; return; }
    if (nodes[p].link == 0) { kind = SAMD_FK_ROOTCHILD; ref = (unsigned)rctok[p]; len = 0u; return; }
    const unsigned L = (unsigned)(nodes[p].length & SAMD_LEN_MASK);
    if (bref[p] && L < lmax) { kind = SAMD_FK_HUB; ref = bref[p]; len = L; return; }
    kind = SAMD_FK_STATE; ref = (unsigned)p; len = L < lmax ? L : lmax;          // lmax = "read it from the node"
}
// one thread per state: its hot word, and -- for a state that owns a block -- every slot of the block (header everywhere, then the edges)
__global__ __launch_bounds__(256) void k_eb_fill(const SamNode *__restrict__ nodes, const SamEdge *__restrict__ spill, long long n, const uint4 *__restrict__ chain,
                                                 const uint32_t *__restrict__ bref, const int32_t *__restrict__ rctok, int tok_bits,
                                                 uint4 *__restrict__ hot, uint4 *__restrict__ blocks) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int *w = reinterpret_cast<const int *>(nodes + s);
    if (s == 0) { hot[0] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu); return; }
    unsigned kind, ref, len;
    eb_fail(nodes, bref, rctok, s, SAMD_EB_IDX_MASK, kind, ref, len);
    const bool single = (w[1] & SAMD_SINGLE) != 0;
    const unsigned flags = (unsigned)w[1] & (unsigned)(SAMD_SINGLE | SAMD_RUN);
    // (samd_common.h: a branching state's word names its own block -- its fail header rides in that block's slots; a root child's its token)
    hot[s] = make_uint4(kind == SAMD_FK_ROOT ? (unsigned)rctok[s] : (single ? ref : bref[s]), (single ? len : 0u) | (kind << 27) | flags, (unsigned)w[2], (unsigned)w[3]);
    const uint32_t my = bref[s];
    if (!my) return;
    const unsigned tmask = (1u << tok_bits) - 1u, lmax = (1u << (32 - tok_bits)) - 1u;
    eb_fail(nodes, bref, rctok, s, lmax, kind, ref, len);                          // (the slot's length field is narrower than the hot word's)
    const unsigned base = samd_eb_base(my), bmask = samd_eb_mask(my);
    const unsigned x_empty = tmask | (len << tok_bits), y_hdr = kind << SAMD_EB_KIND_SHIFT;
    uint4 *blk = blocks + base;
    for (unsigned k = 0; k <= bmask; k++) blk[k] = make_uint4(x_empty, y_hdr, 0xFFFFFFFFu, ref);
    auto put = [&](int t, int d) {
        if (t < 0 || d < 0 || (unsigned)t >= tmask) return;
        unsigned p = samd_eb_hash(t) & bmask;
        for (unsigned probes = 0; probes <= bmask; probes++) {
            const unsigned cur = blk[p].x & tmask;
            if (cur == (unsigned)t) return;                                         // (the spill head repeats ranks 5..7)
            if (cur == tmask) {
                const unsigned dref = bref[d];
                unsigned y = ((unsigned)d & SAMD_EB_IDX_MASK) | y_hdr | (dref ? SAMD_EB_HUB : 0u) | (nodes[d].link == 0 && d > 0 ? SAMD_EB_ROOTCHILD : 0u);
                y |= blk[p].y & SAMD_EB_DISPLACED;                                    // (a bit an earlier, displaced key of this home slot set)
                blk[p] = make_uint4((unsigned)t | (len << tok_bits), y, dref ? dref : chain[d].x, ref);
                if (probes) blk[samd_eb_hash(t) & bmask].y |= SAMD_EB_DISPLACED;      // stored away from its home slot: the home slot says so
                return;
            }
            p = (p + 1) & bmask;
        }
    };
    const int deg = w[5];
    for (int k = 0; k < SAMD_INLINE_EDGES; k++) put(w[SAMD_EDGE_WORD(k)], w[SAMD_EDGE_WORD(k) + 1]);
    if (deg > SAMD_INLINE_EDGES) {
        const SamEdge *sp = spill + w[14];
        const uint32_t slots = samd_spill_slots(deg);
        for (uint32_t k = 0; k < SAMD_SPILL_HEAD + slots; k++) put(sp[k].tok, sp[k].dst);
    }
}

static long long table_budget_bytes() {
    long long budget = 8ll << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { if ((long long)(free_b / 8) < budget) budget = (long long)(free_b / 8); } else (void)hipGetLastError();
    return budget;
}
static int table_slots_per_entry(int arg) {
    static const int per_env = [] { const char *e = getenv("SAMD_BIGRAM_SLOTS_PER_PAIR"); const int v = e ? atoi(e) : 4; return v < 2 ? 2 : (v > 64 ? 64 : v); }();
    return arg > 0 ? (arg < 2 ? 2 : (arg > 64 ? 64 : arg)) : per_env;
}

// the edge table of the branching states; sized and budgeted like the bigram table (slots per entry: the same knob).  An accelerator: when
// the device cannot spare it the walks go through the nodes as before.  SAMD_EDGE_TABLE=0 switches it off (A/B).
static int derive_edge_hash(samd_static_t *s, hipStream_t st, int per_arg) {
    const char *env = getenv("SAMD_EDGE_TABLE");                      // read at every derivation: tests upload the same automaton both ways
    const bool enabled = !(env && env[0] == '0');
    if (s->d_ehash) { (void)hipFree(s->d_ehash); s->d_ehash = nullptr; }
    s->n_ehash = 0;
    if (!enabled || !s->d_chain || s->n_states < 2) return SAMD_OK;
    const long long n = (long long)s->n_states;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    unsigned long long *d_total = nullptr, total = 0;
    if (hipMalloc((void **)&d_total, 8) != hipSuccess) { (void)hipGetLastError(); return SAMD_OK; }
    int rc = SAMD_OK;
    if (hipMemsetAsync(d_total, 0, 8, st) != hipSuccess) rc = SAMD_E_HIP;
    if (rc == SAMD_OK) {
        hipLaunchKernelGGL(k_eh_count, dim3(blocks), dim3(256), 0, st, s->d_nodes, n, d_total);
        if (hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
    }
    (void)hipFree(d_total);
    if (rc != SAMD_OK) { samd_set_error("edge table derivation failed"); return rc; }
    if (total == 0) return SAMD_OK;
    const int per = table_slots_per_entry(per_arg);
    const long long budget = table_budget_bytes();
    long long slots = 1024;
    while (slots < per * (long long)total) slots <<= 1;
    while (slots * 16 > budget && slots >= 4 * (long long)total) slots >>= 1;
    if (slots > (1ll << 31) || slots * 16 > budget) return SAMD_OK;                     // does not fit: go without
    while (hipMalloc(&s->d_ehash, (size_t)slots * 16) != hipSuccess) {
        s->d_ehash = nullptr; (void)hipGetLastError();
        if (slots < 4 * (long long)total || slots <= 1024) return SAMD_OK;
        slots >>= 1;
    }
    if (hipMemsetAsync(s->d_ehash, 0xFF, (size_t)slots * 16, st) != hipSuccess) rc = SAMD_E_HIP;
    if (rc == SAMD_OK) {
        hipLaunchKernelGGL(k_eh_fill, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_spill, n, (const uint4 *)s->d_chain, (uint4 *)s->d_ehash, (uint32_t)(slots - 1));
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
    }
    if (rc != SAMD_OK) { (void)hipFree(s->d_ehash); s->d_ehash = nullptr; samd_set_error("edge table derivation failed"); return rc; }
    s->n_ehash = slots;
    return SAMD_OK;
}

// hot words + edge blocks (samd_common.h, round 6).  On success *out_bref = the per-state block references (device, owned by the caller: the
// bigram table's fill needs them, then they go).  An accelerator like the tables it replaces: whenever something does not fit -- memory, a
// length or an index beyond 27 bits, more than 2^27 slots -- the handle simply has none and the walks use the edge table.  SAMD_EDGE_BLOCKS=0
// switches it off (A/B, tests).
static int derive_edge_blocks(samd_static_t *s, hipStream_t st, int per_arg, uint32_t **out_bref) {
    *out_bref = nullptr;
    const char *env = getenv("SAMD_EDGE_BLOCKS");                     // read at every derivation: tests upload the same automaton every way
    if (s->d_hot) { (void)hipFree(s->d_hot); s->d_hot = nullptr; }
    if (s->d_blocks) { (void)hipFree(s->d_blocks); s->d_blocks = nullptr; }
    s->n_block_slots = s->n_block_states = 0;
    const long long n = (long long)s->n_states;
    const int tok_bits = samd_eb_tok_bits(s->vocab);
    if ((env && env[0] == '0') || !s->d_chain || n < 2 || n > (long long)SAMD_EB_IDX_MASK || tok_bits > 24 || s->vocab < 1 || s->vocab > (1 << 24)) return SAMD_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    uint32_t *d_sizes = nullptr, *d_bref = nullptr; int32_t *d_rctok = nullptr; unsigned long long *d_total = nullptr; unsigned *d_bad = nullptr; void *d_tmp = nullptr;
    void *d_hot = nullptr, *d_blk = nullptr;
    auto cleanup = [&](bool keep_bref) {
        if (d_sizes) (void)hipFree(d_sizes); if (d_rctok) (void)hipFree(d_rctok); if (d_total) (void)hipFree(d_total);
        if (d_bad) (void)hipFree(d_bad); if (d_tmp) (void)hipFree(d_tmp);
        if (!keep_bref && d_bref) { (void)hipFree(d_bref); d_bref = nullptr; }
    };
    auto none = [&]() { (void)hipGetLastError(); cleanup(false); if (d_hot) (void)hipFree(d_hot); if (d_blk) (void)hipFree(d_blk); return SAMD_OK; };
    if (hipMalloc((void **)&d_sizes, (size_t)n * 4) != hipSuccess || hipMalloc((void **)&d_bref, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void **)&d_rctok, (size_t)n * 4) != hipSuccess || hipMalloc((void **)&d_total, 16) != hipSuccess || hipMalloc((void **)&d_bad, 4) != hipSuccess) return none();
    int per = table_slots_per_entry(per_arg);
    const long long budget = table_budget_bytes();
    unsigned long long total[2] = {0, 0};
    for (;;) {
        unsigned bad = 0;
        if (hipMemsetAsync(d_total, 0, 16, st) != hipSuccess || hipMemsetAsync(d_bad, 0, 4, st) != hipSuccess) return none();
        hipLaunchKernelGGL(k_eb_size, dim3(blocks), dim3(256), 0, st, s->d_nodes, n, per, d_sizes, d_total, d_bad);
        if (hipMemcpyAsync(total, d_total, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return none();
        if (bad) return none();
        if (((long long)total[0] * 16 > budget || total[0] > (4ull * SAMD_EB_IDX_MASK)) && per > 2) { per = per / 2 < 2 ? 2 : per / 2; continue; }
        break;
    }
    if ((long long)total[0] * 16 > budget || total[0] > (4ull * SAMD_EB_IDX_MASK)) return none();
    size_t tmp_bytes = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sizes, d_bref, (int)n, st) != hipSuccess) return none();
    if (hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16) != hipSuccess) return none();
    if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sizes, d_bref, (int)n, st) != hipSuccess) return none();
    hipLaunchKernelGGL(k_eb_ref, dim3(blocks), dim3(256), 0, st, d_sizes, d_bref, n);
    if (hipMemsetAsync(d_rctok, 0xFF, (size_t)n * 4, st) != hipSuccess) return none();
    hipLaunchKernelGGL(k_eb_rctok, dim3((unsigned)((s->vocab + 255) / 256)), dim3(256), 0, st, s->d_root, (int)s->vocab, d_rctok);
    if (hipMalloc(&d_hot, (size_t)n * 16) != hipSuccess || hipMalloc(&d_blk, (size_t)(total[0] ? total[0] : 1) * 16) != hipSuccess) return none();
    hipLaunchKernelGGL(k_eb_fill, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_spill, n, (const uint4 *)s->d_chain, d_bref, d_rctok, tok_bits, (uint4 *)d_hot, (uint4 *)d_blk);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError(); cleanup(false); (void)hipFree(d_hot); (void)hipFree(d_blk);
        samd_set_error("edge block derivation failed"); return SAMD_E_HIP;
    }
    s->d_hot = d_hot; s->d_blocks = d_blk; s->n_block_slots = (int64_t)total[0]; s->n_block_states = (int64_t)total[1];
    cleanup(true);
    *out_bref = d_bref;
    return SAMD_OK;
}

static int derive_topk_counts(samd_static_t *s, hipStream_t st) {
    static const bool enabled = [] { const char *e = getenv("SAMD_TOPK_COUNTS"); return !(e && e[0] == '0'); }();       // A/B switch, read once
    if (!enabled || s->kind != SAMD_KIND_COUNT) return SAMD_OK;
    const long long n = (long long)s->n_states * SAMD_TOPK;
    if (!s->d_topk_cnt && hipMalloc(&s->d_topk_cnt, (size_t)n * 4) != hipSuccess) { s->d_topk_cnt = nullptr; samd_set_error("hipMalloc(top-k counts) failed"); return SAMD_E_HIP; }
    hipLaunchKernelGGL(k_topk_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->d_nodes, s->d_spill, (long long)s->n_states, (int32_t *)s->d_topk_cnt);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { samd_set_error("top-k count derivation failed"); return SAMD_E_HIP; }
    return SAMD_OK;
}

static int derive_root_hash(samd_static_t *s, hipStream_t st, int per_pair_arg = 0, const uint32_t *d_bref = nullptr) {
    static const bool enabled = [] { const char *e = getenv("SAMD_ROOT_HASH"); return !(e && e[0] == '0'); }();      // A/B switch, read once
    if (s->d_root16) { (void)hipFree(s->d_root16); s->d_root16 = nullptr; }
    if (s->d_d1hash) { (void)hipFree(s->d_d1hash); s->d_d1hash = nullptr; }
    if (s->d_rc_bits) { (void)hipFree(s->d_rc_bits); s->d_rc_bits = nullptr; }
    s->n_d1hash = 0;
    if (!enabled || s->vocab < 1 || s->vocab > (1 << 24) || !s->d_chain) return SAMD_OK;
    const int vocab = (int)s->vocab;
    const unsigned blocks = (unsigned)((vocab + 255) / 256);
    unsigned long long *d_total = nullptr, total = 0;
    int rc = SAMD_OK;
    if (hipMalloc((void **)&d_total, 8) != hipSuccess) { samd_set_error("hipMalloc(bigram count) failed"); return SAMD_E_HIP; }
    if (hipMemsetAsync(d_total, 0, 8, st) != hipSuccess) rc = SAMD_E_HIP;
    if (rc == SAMD_OK) {
        hipLaunchKernelGGL(k_bg_count, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_root, vocab, d_total);
        if (hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
    }
    (void)hipFree(d_total);
    // slots per pair: a lock-step wave of the BATCHED walk pays a second probe round whenever ANY of its 64 lanes collides, so that launch
    // likes a very sparse table -- measured on the bench automaton (4.5 M pairs, same box): >= 2 x pairs 0.188 ms per launch, 4 x 0.176,
    // 8 x 0.170, 16 x 0.166 (2 GB).  The product path (st_transfer_tokens: ONE cursor, uniform addresses) gains nothing from the sparsity, and
    // the table is per GPU replica next to weights and KV cache, so the DEFAULT is 4 slots per pair (round 5; rounds 3-4: 16) and a caller that
    // runs the batched walk asks for more: samd_static_set_bigram_slots() / SAMD_BIGRAM_SLOTS_PER_PAIR (2 .. 64).  Whatever is asked for, the
    // table stays under 8 GB and under an eighth of the device memory free right now, as long as 2 x pairs fit in that.
    static const int per_pair_env = [] { const char *e = getenv("SAMD_BIGRAM_SLOTS_PER_PAIR"); const int v = e ? atoi(e) : 4; return v < 2 ? 2 : (v > 64 ? 64 : v); }();
    const int per_pair = per_pair_arg > 0 ? (per_pair_arg < 2 ? 2 : (per_pair_arg > 64 ? 64 : per_pair_arg)) : per_pair_env;
    long long budget = 8ll << 30;
    { size_t free_b = 0, total_b = 0; if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (long long)(free_b / 8) < budget) budget = (long long)(free_b / 8); else (void)hipGetLastError(); }
    long long slots = 1024;
    while (slots < per_pair * (long long)total) slots <<= 1;                       // load factor in (1 / 2 per_pair, 1 / per_pair]
    while (slots * 16 > budget && slots >= 4 * (long long)total) slots >>= 1;
    if (rc == SAMD_OK && slots > (1ll << 31)) rc = -1;                            // the mask does not fit 32 bits: go without the table
    const size_t bit_bytes = (size_t)((vocab + 31) / 32) * 4;
    if (rc == SAMD_OK) {
        // the table is an accelerator, not part of the image: when the device cannot spare its preferred size, halve it down to 2 slots per pair,
        // and when even that does not fit go without it (walks then resolve root children through their nodes)
        while (hipMalloc(&s->d_d1hash, (size_t)slots * 16) != hipSuccess) {
            s->d_d1hash = nullptr; (void)hipGetLastError();
            if (slots < 4 * (long long)total || slots <= 1024) { rc = -1; break; }
            slots >>= 1;
        }
        if (rc == SAMD_OK && (hipMalloc(&s->d_root16, (size_t)vocab * 16) != hipSuccess || hipMalloc(&s->d_rc_bits, bit_bytes) != hipSuccess)) { (void)hipGetLastError(); rc = -1; }
        if (rc == SAMD_OK && (hipMemsetAsync(s->d_d1hash, 0xFF, (size_t)slots * 16, st) != hipSuccess || hipMemsetAsync(s->d_rc_bits, 0, bit_bytes, st) != hipSuccess)) rc = SAMD_E_HIP;
        if (rc == SAMD_OK) {
            hipLaunchKernelGGL(k_bg_fill, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_spill, s->d_root, vocab, (const uint4 *)s->d_chain, (uint4 *)s->d_root16,
                               (uint4 *)s->d_d1hash, (uint32_t)(slots - 1), (uint32_t *)s->d_rc_bits, samd_chain_w(vocab), s->d_ehash ? 1 : 0, d_bref);
            // (only the block path reads the bit, and only a handle with blocks may carry it: the edge-table path takes the dst word's low 31 bits as the index)
            if (d_bref) hipLaunchKernelGGL(k_bg_displaced, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, (uint4 *)s->d_d1hash, (uint32_t)(slots - 1), samd_chain_w(vocab));
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
        }
        if (rc == SAMD_OK) s->n_d1hash = slots;
    }
    if (rc != SAMD_OK) {
        if (s->d_root16) { (void)hipFree(s->d_root16); s->d_root16 = nullptr; }
        if (s->d_d1hash) { (void)hipFree(s->d_d1hash); s->d_d1hash = nullptr; }
        if (s->d_rc_bits) { (void)hipFree(s->d_rc_bits); s->d_rc_bits = nullptr; }
        s->n_d1hash = 0;
        if (rc == SAMD_E_HIP) { samd_set_error("bigram table derivation failed"); return rc; }
    }
    return SAMD_OK;
}

// the derived walk tables in the order they depend on each other: edge blocks (+ hot words) when they fit, else the edge table; then the
// bigram table, whose entries name a target's block.  The block path needs the bigram table: without one the blocks go and the edge table
// (which does not) comes back.
static int derive_walk_tables(samd_static_t *s, hipStream_t st, int per_arg) {
    uint32_t *d_bref = nullptr;
    int rc = derive_edge_blocks(s, st, per_arg, &d_bref);
    if (rc == SAMD_OK && !s->d_blocks) rc = derive_edge_hash(s, st, per_arg);
    else if (rc == SAMD_OK && s->d_ehash) { (void)hipFree(s->d_ehash); s->d_ehash = nullptr; s->n_ehash = 0; }
    if (rc == SAMD_OK) rc = derive_root_hash(s, st, per_arg, d_bref);
    if (d_bref) (void)hipFree(d_bref);
    if (rc == SAMD_OK && s->d_blocks && !s->d_d1hash) {
        (void)hipFree(s->d_hot); (void)hipFree(s->d_blocks); s->d_hot = s->d_blocks = nullptr; s->n_block_slots = s->n_block_states = 0;
        rc = derive_edge_hash(s, st, per_arg);
        if (rc == SAMD_OK) rc = derive_root_hash(s, st, per_arg, nullptr);
    }
    return rc;
}

int samd_static_derive_chain(samd_static_t *s, void *stream) {
    if (!s || !s->uploaded || !s->d_nodes) return SAMD_E_INVALID;
    if (!s->d_chain && hipMalloc(&s->d_chain, (size_t)s->n_states * 16) != hipSuccess) { s->d_chain = nullptr; samd_set_error("hipMalloc(chain words) failed"); return SAMD_E_HIP; }
    const unsigned blocks = (unsigned)((s->n_states + 255) / 256);
    if (samd_chain_w(s->vocab) == 8) hipLaunchKernelGGL(k_build_chain<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->d_nodes, (long long)s->n_states, (uint4 *)s->d_chain);
    else hipLaunchKernelGGL(k_build_chain<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->d_nodes, (long long)s->n_states, (uint4 *)s->d_chain);
    LAUNCHCHK();
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { samd_set_error("chain-word derivation failed"); return SAMD_E_HIP; }
    const int rc = derive_walk_tables(s, (hipStream_t)stream, 0);
    return rc != SAMD_OK ? rc : derive_topk_counts(s, (hipStream_t)stream);
}

int samd_static_set_bigram_slots(samd_static_t *s, int32_t slots_per_pair, void *stream) {
    if (!s || !s->uploaded || !s->d_nodes || slots_per_pair < 0) { samd_set_error("samd_static_set_bigram_slots: invalid argument"); return SAMD_E_INVALID; }
    if (!s->d_chain) return SAMD_OK;                                           // no derived tables on this handle: nothing to re-size
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SAMD_E_HIP;      // nothing may still read the tables that are replaced
    return derive_walk_tables(s, (hipStream_t)stream, slots_per_pair);                   // (every table follows the same knob)
}

int samd_static_walk(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                     int32_t commit, int32_t *d_trace, void *stream) {
    if (!sam || !sam->uploaded || B < 0 || T < 0) { samd_set_error("samd_static_walk: invalid argument"); return SAMD_E_INVALID; }
    if (B == 0 || T == 0) return SAMD_OK;                 // empty batch / no tokens: nothing to do
    if (!d_cursors || !d_tokens) { samd_set_error("samd_static_walk: null pointer"); return SAMD_E_INVALID; }
    const int threads = 256, blocks = (B + threads - 1) / threads;
    launch_walk(sam, blocks, threads, (hipStream_t)stream, d_cursors, commit ? d_cursors : nullptr, d_tokens, B, T, d_trace, (unsigned long long *)nullptr);
    LAUNCHCHK();
    return SAMD_OK;
}

// lookup over B cursors: the walk's result goes to d_out, the cursors stay (static_sam.py:122-125 returns the pair and commits nothing)
int samd_static_lookup_batch(const samd_static_t *sam, const int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t *d_out, uint64_t *d_visited, void *stream) {
    if (!sam || !sam->uploaded || B < 0 || T < 0) { samd_set_error("samd_static_lookup_batch: invalid argument"); return SAMD_E_INVALID; }
    if (B == 0) return SAMD_OK;
    if (!d_cursors || !d_out || (T > 0 && !d_tokens)) { samd_set_error("samd_static_lookup_batch: null pointer"); return SAMD_E_INVALID; }
    if (T == 0) {                                          // no token: the result is the cursor itself
        if (d_out != d_cursors) HIPCHK(hipMemcpyAsync(d_out, d_cursors, (size_t)B * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return SAMD_OK;
    }
    const int threads = 256, blocks = (B + threads - 1) / threads;
    launch_walk(sam, blocks, threads, (hipStream_t)stream, d_cursors, d_out, d_tokens, B, T, (int32_t *)nullptr, (unsigned long long *)d_visited);
    LAUNCHCHK();
    return SAMD_OK;
}

// same launch, additionally accumulating the number of visited states into *d_visited (u64, device);
// used by bench.py to turn kernel time into algorithmic bytes (16 B per visited state).
int samd_static_walk_counted(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t commit, uint64_t *d_visited, void *stream) {
    if (!sam || !sam->uploaded || !d_cursors || !d_tokens || B <= 0 || T <= 0 || !d_visited) return SAMD_E_INVALID;
    const int threads = 256, blocks = (B + threads - 1) / threads;
    launch_walk(sam, blocks, threads, (hipStream_t)stream, d_cursors, commit ? d_cursors : nullptr, d_tokens, B, T, (int32_t *)nullptr, (unsigned long long *)d_visited);
    LAUNCHCHK();
    return SAMD_OK;
}

static int walk_streams(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T, int32_t commit,
                        int32_t *d_trace, uint64_t *d_visited, void *stream) {
    if (!sam || !sam->uploaded || B < 0 || T < 0) { samd_set_error("samd_static_walk_streams: invalid argument"); return SAMD_E_INVALID; }
    if (B == 0 || T == 0) return SAMD_OK;
    if (!d_cursors || !d_tokens) { samd_set_error("samd_static_walk_streams: null pointer"); return SAMD_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    int32_t *tm = nullptr, *tr = nullptr;
    const size_t n = (size_t)B * T;
    if (hipMallocAsync((void **)&tm, n * 4, st) != hipSuccess) { samd_set_error("samd_static_walk_streams: scratch allocation failed"); return SAMD_E_HIP; }
    if (d_trace && hipMallocAsync((void **)&tr, n * 8, st) != hipSuccess) { (void)hipFreeAsync(tm, st); samd_set_error("samd_static_walk_streams: scratch allocation failed"); return SAMD_E_HIP; }
    hipLaunchKernelGGL(k_transpose_i32<1>, dim3((T + 31) / 32, (B + 31) / 32), dim3(256), 0, st, d_tokens, tm, B, T);
    launch_walk(sam, (B + 255) / 256, 256, st, d_cursors, commit ? d_cursors : nullptr, tm, B, T, tr, (unsigned long long *)d_visited);
    if (d_trace) hipLaunchKernelGGL(k_transpose_i32<2>, dim3((B + 31) / 32, (T + 31) / 32), dim3(256), 0, st, tr, d_trace, T, B);
    (void)hipFreeAsync(tm, st);
    if (tr) (void)hipFreeAsync(tr, st);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_static_walk_streams(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t commit, int32_t *d_trace, void *stream) {
    return walk_streams(sam, d_cursors, d_tokens, B, T, commit, d_trace, nullptr, stream);
}

int samd_static_walk_streams_counted(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                                     int32_t commit, uint64_t *d_visited, void *stream) {
    if (!d_visited) return SAMD_E_INVALID;
    return walk_streams(sam, d_cursors, d_tokens, B, T, commit, nullptr, d_visited, stream);
}

int samd_session_create(int32_t max_tokens, samd_session_t **out) {
    if (!out || max_tokens < 1 || max_tokens > (1 << 24)) { samd_set_error("samd_session_create: invalid max_tokens"); return SAMD_E_INVALID; }
    if (samd_device_count() < 1) { samd_set_error("no HIP device"); return SAMD_E_NODEVICE; }
    samd_session_t *s = (samd_session_t *)calloc(1, sizeof(samd_session_t));
    if (!s) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    s->max_tokens = max_tokens;
    SessionDev &D = s->dev;
    D.max_tokens = max_tokens;
    D.cap_states = 2 * max_tokens + 2;
    D.cap_text = max_tokens + 2;
    uint32_t H = 1024; while (H < (uint32_t)max_tokens * 8u) H <<= 1;
    D.hmask = H - 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // report block: dmeta | verdict | acc_tokens | kv_index | counters | meta are contiguous so that one
    // D2H copy (samd_session_report_async) hands the host everything it polls per step
    const size_t o_link = carve(4ull * D.cap_states), o_len = carve(4ull * D.cap_states), o_me = carve(4ull * D.cap_states),
                 o_head = carve(4ull * D.cap_states), o_tail = carve(4ull * D.cap_states), o_hk = carve(8ull * H),
                 o_hd = carve(4ull * H), o_hn = carve(4ull * H), o_text = carve(4ull * D.cap_text),
                 o_tok = carve(4 * SAMD_MAX_DRAFT), o_par = carve(4 * SAMD_MAX_DRAFT), o_pos = carve(4 * SAMD_MAX_DRAFT), o_mask = carve(8 * 2 * SAMD_MAX_DRAFT),
                 o_ret = carve(4 * SAMD_MAX_DRAFT * SAMD_MAX_DRAFT),
                 o_rep = carve(4 * SAMD_REPORT_INTS), o_st = carve(4), o_cl = carve(4), o_seq = carve(4);
    s->arena_bytes = off;
    if (hipMalloc(&s->arena, off) != hipSuccess) { free(s); samd_set_error("hipMalloc(session arena) failed"); return SAMD_E_HIP; }
    char *base = (char *)s->arena;
    D.link = (int32_t *)(base + o_link); D.length = (int32_t *)(base + o_len); D.minend = (int32_t *)(base + o_me);
    D.head = (int32_t *)(base + o_head); D.tail = (int32_t *)(base + o_tail); D.hkey = (uint64_t *)(base + o_hk);
    D.hdst = (int32_t *)(base + o_hd); D.hnext = (int32_t *)(base + o_hn); D.text = (int32_t *)(base + o_text);
    D.tokens = (int32_t *)(base + o_tok); D.parent = (int32_t *)(base + o_par);
    D.position = (int32_t *)(base + o_pos); D.mask = (uint64_t *)(base + o_mask); D.mask_hi = D.mask + SAMD_MAX_DRAFT; D.retrieve = (int32_t *)(base + o_ret);
    int32_t *rep = (int32_t *)(base + o_rep);
    D.dmeta = rep + SAMD_REP_DMETA; D.verdict = rep + SAMD_REP_VERDICT; D.acc_tokens = rep + SAMD_REP_TOKENS;
    D.kv_index = rep + SAMD_REP_KVINDEX; D.counters = rep + SAMD_REP_COUNTERS; D.meta = rep + SAMD_REP_META;
    D.start_token = (int32_t *)(base + o_st); D.cache_length = (int32_t *)(base + o_cl);
    D.push_seq = (int32_t *)(base + o_seq); D.h_report = nullptr;
    if (hipMemset(s->arena, 0, off) != hipSuccess) { (void)hipFree(s->arena); free(s); return SAMD_E_HIP; }
    *out = s;
    // the report target exists from the start: SessionDev travels BY VALUE into every launch (and into every captured graph), so a target
    // created later would be missing from the steps captured before it (they would never push)
    int32_t *target = nullptr;
    int rc = samd_session_report_target(s, &target);
    if (rc == SAMD_OK) rc = samd_session_reset(s, nullptr);
    if (rc == SAMD_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = SAMD_E_HIP;
    if (rc) { samd_session_free(s); *out = nullptr; }
    return rc;
}

void samd_session_free(samd_session_t *s) {
    if (!s) return;
    if (s->arena) (void)hipFree(s->arena);
    if (s->h_report_host) (void)hipHostFree(s->h_report_host);
    free(s);
}

int samd_session_report_target(samd_session_t *s, int32_t **out_host) {
    if (!s || !out_host) return SAMD_E_INVALID;
    if (!s->h_report_host) {
        void *h = nullptr, *d = nullptr;
        const size_t bytes = 4 * (SAMD_REPORT_INTS + 16);
        if (hipHostMalloc(&h, bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { samd_set_error("hipHostMalloc(report target) failed"); return SAMD_E_HIP; }
        memset(h, 0, bytes);
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); samd_set_error("hipHostGetDevicePointer(report target) failed"); return SAMD_E_HIP; }
        s->h_report_host = (int32_t *)h;
        s->dev.h_report = (int32_t *)d;
    }
    *out_host = s->h_report_host;
    return SAMD_OK;
}

int samd_report_wait(const int32_t *h_report, int32_t last_seq, int64_t timeout_us) {
    if (!h_report) return SAMD_E_INVALID;
    const volatile int32_t *seq = h_report + SAMD_REPORT_INTS;
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spin = 0;; spin++) {
        if (*seq != last_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return SAMD_OK; }
        __builtin_ia32_pause();
        if ((spin & 1023u) == 1023u) {
            timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
            if ((t.tv_sec - t0.tv_sec) * 1000000ll + (t.tv_nsec - t0.tv_nsec) / 1000 > timeout_us) return SAMD_E_CAPACITY;
        }
    }
}

static int launch_session(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, StepArgs &A, void *stream) {
    if (!s) { samd_set_error("null session"); return SAMD_E_INVALID; }
    if (sam && !sam->uploaded) { samd_set_error("static automaton not uploaded"); return SAMD_E_INVALID; }
    samd_params_t P; memset(&P, 0, sizeof(P));
    if (p) P = *p;
    if (p && (P.max_predicts > SAMD_MAX_DRAFT || P.n_predicts > SAMD_MAX_DRAFT || P.max_predicts < 0 || P.n_predicts < 0)) {
        samd_set_error("max_predicts / n_predicts must be in [0, %d]", SAMD_MAX_DRAFT); return SAMD_E_INVALID;
    }
    A.have_static = sam ? 1 : 0;
    hipLaunchKernelGGL(k_session, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, s->dev, static_view(sam), P, A);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_session_reset(samd_session_t *s, void *stream) {
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_RESET;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_dyn_add_tokens(samd_session_t *s, const int32_t *d_tokens, int32_t n, const int32_t *d_n, void *stream) {
    if (n < 0 || (!d_tokens && n > 0)) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_ADD; A.tokens = d_tokens; A.n = n; A.d_n = d_n;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_dyn_walk(samd_session_t *s, const int32_t *d_tokens, int32_t n, int32_t commit, int32_t *d_out, void *stream) {
    if (n < 0 || (!d_tokens && n > 0)) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_DYN_WALK; A.tokens = d_tokens; A.n = n; A.commit = commit; A.out2 = d_out;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_session_static_walk(samd_session_t *s, const samd_static_t *sam, const int32_t *d_tokens, int32_t n,
                             const int32_t *d_n, int32_t commit, int32_t *d_out, void *stream) {
    if (n < 0 || (!d_tokens && n > 0)) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_ST_WALK; A.tokens = d_tokens; A.n = n; A.d_n = d_n; A.commit = commit; A.out2 = d_out;
    return launch_session(s, sam, nullptr, A, stream);
}

int samd_session_set_cursors(samd_session_t *s, int32_t di, int32_t dl, int32_t si, int32_t sl, void *stream) {
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_SET_CURSORS; A.c0 = di; A.c1 = dl; A.c2 = si; A.c3 = sl;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_session_draft(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, const int32_t *d_start_token, void *stream) {
    if (!p || !d_start_token) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_DRAFT; A.start_token = d_start_token;
    return launch_session(s, sam, p, A, stream);
}

int samd_session_draft_seq(samd_session_t *s, const samd_params_t *p, int32_t index, int32_t match, int32_t start_token, void *stream) {
    if (!p || index < 0) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_DRAFT_SEQ; A.index = index; A.match = match; A.start = start_token;
    return launch_session(s, nullptr, p, A, stream);
}

int samd_session_draft_tree(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, int32_t index, int32_t match,
                            int32_t start_token, void *stream) {
    if (!p || !sam || index < 0 || index >= sam->n_states) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_DRAFT_TREE; A.index = index; A.match = match; A.start = start_token;
    return launch_session(s, sam, p, A, stream);
}

int samd_session_draft_fixed(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, int32_t source, int32_t index,
                             int32_t start_token, void *stream) {
    if (!p || index < 0 || (source == 1 && (!sam || sam->kind != SAMD_KIND_ENDPOS || index >= sam->n_states))) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_DRAFT_FIXED; A.index = index; A.start = start_token; A.source = source;
    return launch_session(s, sam, p, A, stream);
}

int samd_session_set_draft(samd_session_t *s, const int32_t *d_tokens, const int32_t *d_parent, int32_t n, int32_t type, void *stream) {
    if (!d_tokens || !d_parent || n < 1 || n > SAMD_MAX_DRAFT) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_SET_DRAFT; A.tokens = d_tokens; A.parents = d_parent; A.n = n;
    A.type = type & 0xff; A.reverse = (type >> 8) & 1;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_session_set_draft_if_deferred(samd_session_t *s, const int32_t *d_tokens, const int32_t *d_parent, int32_t n,
                                       int32_t reverse_leaves, void *stream) {
    if (!d_tokens || !d_parent || n < 1 || n > SAMD_MAX_DRAFT) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_SET_DRAFT; A.tokens = d_tokens; A.parents = d_parent; A.n = n;
    A.type = 1; A.reverse = reverse_leaves ? 1 : 0; A.only_if_deferred = 1;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_session_set_start_token(samd_session_t *s, const int32_t *d_src, void *stream) {
    if (!s || !d_src) return SAMD_E_INVALID;
    HIPCHK(hipMemcpyAsync(s->dev.start_token, d_src, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SAMD_OK;
}

int samd_session_report_async(samd_session_t *s, int32_t *h_dst, void *stream) {
    if (!s || !h_dst) return SAMD_E_INVALID;
    HIPCHK(hipMemcpyAsync(h_dst, s->dev.dmeta, 4 * SAMD_REPORT_INTS, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SAMD_OK;
}

int samd_session_accept(samd_session_t *s, const int32_t *d_node_argmax, void *stream) {
    if (!d_node_argmax) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_ACCEPT; A.node_argmax = d_node_argmax;
    return launch_session(s, nullptr, nullptr, A, stream);
}

int samd_session_commit(samd_session_t *s, const samd_static_t *sam, void *stream) {
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_COMMIT;
    return launch_session(s, sam, nullptr, A, stream);
}

int samd_session_step(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, const int32_t *d_node_argmax, void *stream) {
    if (!p || !d_node_argmax || !s) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_ACCEPT | OP_COMMIT | OP_DRAFT; A.node_argmax = d_node_argmax;
    A.start_token = s->dev.start_token;
    A.push_report = 1;                                    // (no-op without samd_session_report_target)
    return launch_session(s, sam, p, A, stream);
}

int samd_session_step_given(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, const int32_t *d_best_accept, const int32_t *d_next_token,
                            void *stream) {
    if (!p || !d_best_accept || !d_next_token || !s) return SAMD_E_INVALID;
    StepArgs A; memset(&A, 0, sizeof(A)); A.ops = OP_ACCEPT_GIVEN | OP_COMMIT | OP_DRAFT; A.given = d_best_accept; A.given_next = d_next_token;
    A.start_token = s->dev.start_token;
    return launch_session(s, sam, p, A, stream);
}

int samd_session_candidates(samd_session_t *s, int64_t *d_candidates, int32_t *d_rowmap, int32_t capacity, void *stream) {
    if (!s || !d_candidates || !d_rowmap || capacity < 1) return SAMD_E_INVALID;
    hipLaunchKernelGGL(k_session_candidates, dim3(1), dim3(256), 0, (hipStream_t)stream, s->dev, (long long *)d_candidates, d_rowmap, capacity);
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_session_device_views(samd_session_t *s, void *out[16]) {
    if (!s || !out) return SAMD_E_INVALID;
    const SessionDev &D = s->dev;
    out[0] = D.tokens; out[1] = D.parent; out[2] = D.position; out[3] = D.mask; out[4] = D.retrieve; out[5] = D.dmeta;
    out[6] = D.verdict; out[7] = D.acc_tokens; out[8] = D.kv_index; out[9] = D.start_token; out[10] = D.cache_length;
    out[11] = D.text + 1; out[12] = D.counters; out[13] = D.meta; out[14] = nullptr; out[15] = nullptr;
    return SAMD_OK;
}

#define D2H(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream))

int samd_session_read_draft(samd_session_t *s, samd_draft_host_t *out, void *stream) {
    if (!s || !out) return SAMD_E_INVALID;
    const SessionDev &D = s->dev;
    int32_t dm[D_COUNT];
    D2H(dm, D.dmeta, sizeof(dm));
    D2H(out->tokens, D.tokens, 4 * SAMD_MAX_DRAFT); D2H(out->parent, D.parent, 4 * SAMD_MAX_DRAFT); D2H(out->position, D.position, 4 * SAMD_MAX_DRAFT);
    D2H(out->mask, D.mask, 8 * SAMD_MAX_DRAFT); D2H(out->mask_hi, D.mask_hi, 8 * SAMD_MAX_DRAFT); D2H(out->retrieve, D.retrieve, 4 * SAMD_MAX_DRAFT * SAMD_MAX_DRAFT);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    out->type = dm[D_TYPE]; out->n = dm[D_N]; out->n_leaves = dm[D_NLEAVES]; out->max_depth = dm[D_MAXDEPTH];
    out->index_dyn = dm[D_IDX_DYN]; out->match_dyn = dm[D_MATCH_DYN]; out->index_static = dm[D_IDX_ST]; out->match_static = dm[D_MATCH_ST];
    return SAMD_OK;
}

int samd_session_read_verdict(samd_session_t *s, samd_verdict_host_t *out, void *stream) {
    if (!s || !out) return SAMD_E_INVALID;
    const SessionDev &D = s->dev;
    int32_t v[V_COUNT];
    D2H(v, D.verdict, sizeof(v)); D2H(out->tokens, D.acc_tokens, 4 * SAMD_MAX_DRAFT); D2H(out->kv_index, D.kv_index, 4 * SAMD_MAX_DRAFT);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    out->best = v[V_BEST]; out->accept = v[V_ACCEPT]; out->next_node = v[V_NEXT_NODE]; out->next_token = v[V_NEXT_TOKEN];
    return SAMD_OK;
}

int samd_session_export(samd_session_t *s, int64_t out_info[10], int32_t *h_link, int32_t *h_length, int32_t *h_minend,
                        int32_t *h_deg, int32_t *h_edge_tok, int32_t *h_edge_dst, int32_t *h_text, void *stream) {
    if (!s || !out_info) return SAMD_E_INVALID;
    const SessionDev &D = s->dev;
    int32_t m[M_COUNT];
    D2H(m, D.meta, sizeof(m));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 10; i++) out_info[i] = m[i];
    const int ns = m[M_NSTATES], nt = m[M_NTEXT];
    if (h_link) D2H(h_link, D.link, 4ull * ns);
    if (h_length) D2H(h_length, D.length, 4ull * ns);
    if (h_minend) D2H(h_minend, D.minend, 4ull * ns);
    if (h_text) D2H(h_text, D.text, 4ull * nt);
    if (h_deg || h_edge_tok || h_edge_dst) {
        const size_t H = (size_t)D.hmask + 1;
        std::vector<uint64_t> hk(H); std::vector<int32_t> hd(H), hn(H), head(ns);
        D2H(hk.data(), D.hkey, 8 * H); D2H(hd.data(), D.hdst, 4 * H); D2H(hn.data(), D.hnext, 4 * H); D2H(head.data(), D.head, 4ull * ns);
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        size_t k = 0;
        for (int st = 0; st < ns; st++) {
            int d = 0;
            for (int e = head[st]; e >= 0; e = hn[e]) {
                if (h_edge_tok) h_edge_tok[k] = (int32_t)(uint32_t)hk[e];
                if (h_edge_dst) h_edge_dst[k] = hd[e];
                k++; d++;
            }
            if (h_deg) h_deg[st] = d;
        }
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return SAMD_OK;
}

int samd_session_set_cache_length(samd_session_t *s, int32_t length, void *stream) {
    if (!s || length < 0) return SAMD_E_INVALID;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)s->dev.cache_length, length, 1, (hipStream_t)stream));
    return SAMD_OK;
}

int samd_session_get_cache_length(samd_session_t *s, int32_t *h_out, void *stream) {
    if (!s || !h_out) return SAMD_E_INVALID;
    D2H(h_out, s->dev.cache_length, 4);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return SAMD_OK;
}

int samd_scripted_argmax(samd_session_t *s, const int32_t *d_target, int32_t n_target, int32_t vocab, int32_t *d_out, void *stream) {
    if (!s || !d_target || !d_out || n_target < 0 || vocab < 4) return SAMD_E_INVALID;
    hipLaunchKernelGGL(k_scripted_argmax, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, s->dev, d_target, n_target, vocab, d_out);
    LAUNCHCHK();
    return SAMD_OK;
}

static int scripted_logits(samd_session_t *s, const int32_t *d_argmax, void *d_logits, int32_t dtype, int32_t rows, int64_t row_stride, int32_t markov_vocab,
                           int order, void *stream) {
    if (!s || !d_argmax || !d_logits || markov_vocab < 4 || row_stride < markov_vocab) return SAMD_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
#define SCRIPTED(T) do { if (order == 2) hipLaunchKernelGGL((k_scripted_logits<T, 2>), dim3(SAMD_MAX_DRAFT), dim3(WAVE), 0, st, s->dev, d_argmax, (T *)d_logits, (long long)row_stride, markov_vocab, rows); \
                         else hipLaunchKernelGGL((k_scripted_logits<T, 1>), dim3(SAMD_MAX_DRAFT), dim3(WAVE), 0, st, s->dev, d_argmax, (T *)d_logits, (long long)row_stride, markov_vocab, rows); } while (0)
    if (dtype == SAMD_F16) SCRIPTED(_Float16);
    else if (dtype == SAMD_BF16) SCRIPTED(__bf16);
    else if (dtype == SAMD_F32) SCRIPTED(float);
    else return SAMD_E_INVALID;
#undef SCRIPTED
    LAUNCHCHK();
    return SAMD_OK;
}

int samd_scripted_logits(samd_session_t *s, const int32_t *d_argmax, void *d_logits, int32_t dtype, int32_t rows, int64_t row_stride, int32_t markov_vocab, void *stream) {
    return scripted_logits(s, d_argmax, d_logits, dtype, rows, row_stride, markov_vocab, 2, stream);
}

int samd_scripted_logits_order1(samd_session_t *s, const int32_t *d_argmax, void *d_logits, int32_t dtype, int32_t rows, int64_t row_stride, int32_t hot_vocab,
                                void *stream) {
    return scripted_logits(s, d_argmax, d_logits, dtype, rows, row_stride, hot_vocab, 1, stream);
}

int samd_tree_buffers(const int32_t *d_parent, int32_t n, int32_t reverse_leaves, int32_t *d_position, uint64_t *d_mask,
                      uint8_t *d_mask_bool, int32_t *d_retrieve, int32_t *d_shape, void *stream) {
    if (!d_parent || n < 1 || n > SAMD_MAX_DRAFT) { samd_set_error("samd_tree_buffers: n must be in [1,128]"); return SAMD_E_INVALID; }
    hipLaunchKernelGGL(k_tree_buffers, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, d_parent, n, reverse_leaves, d_position, d_mask,
                       d_mask_bool, d_retrieve, d_shape);
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
