// sam_kernels.hip -- gfx950 kernels + C ABI of the suffix-automaton draft path.
// Reference functions replaced: see include/samd_hip.h (one citation per entry point).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sam_device.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { samd_set_error("%s: %s", #x, hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { samd_set_error("kernel launch: %s", hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

static inline StaticDev static_view(const samd_static_t *s) {
    StaticDev v;
    if (!s) { memset(&v, 0, sizeof(v)); return v; }
    v.nodes = s->d_nodes; v.root_next = s->d_root; v.spill = s->d_spill; v.text = s->d_text;
    v.n_states = (int32_t)s->n_states; v.vocab = (int32_t)s->vocab; v.n_text = (int32_t)s->n_text; v.kind = s->kind;
    v.chain = (const uint4 *)s->d_chain; v.chain_w = s->vocab < 65535 ? 8 : 4;
    v.root16 = (const uint4 *)s->d_root16; v.d1hash = (const SamEdge *)s->d_d1hash;
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
template <int W, bool CHAIN>
__global__ __launch_bounds__(256) void k_static_walk(StaticDev S, int32_t *__restrict__ cursors,
                                                     const int32_t *__restrict__ tokens, int B, int T, int commit,
                                                     int32_t *__restrict__ trace, unsigned long long *__restrict__ visited_total) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long visited = 0;
    if (b < B) {
        int2 c = reinterpret_cast<const int2 *>(cursors)[b];
        int idx = c.x, len = c.y;
        int tok = tokens[b];
        ChainWord cw = chain_none();
        RootChild rc = rootchild_none();
        for (int t = 0; t < T; t++) {
            const int nxt = (t + 1 < T) ? tokens[(size_t)(t + 1) * B + b] : 0;
            if (CHAIN) visited += st_transfer_chain<W>(S, idx, len, tok, cw, rc);
            else visited += st_transfer(S, idx, len, tok);
            if (trace) reinterpret_cast<int2 *>(trace)[(size_t)t * B + b] = make_int2(idx, len);
            tok = nxt;
        }
        if (commit) reinterpret_cast<int2 *>(cursors)[b] = make_int2(idx, len);
    }
    if (visited_total) {
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if ((threadIdx.x & 63) == 0 && visited) atomicAdd(visited_total, visited);
    }
}

// ================================================================================================
// batched walk, STREAM-MAJOR tokens, lanes decoupled (round 3).  scripts/walk_sched_sim.py replays bench.py's workload through
// st_transfer_chain's rule: a lane needs 21.6 dependent memory rounds for its 16 tokens on average, but a wave of k_static_walk
// (lock-step: at every token all 64 lanes wait for the slowest) spends 82 -- and even fully decoupled lanes leave the wave waiting
// for its slowest LANE (49 rounds).  So here (a) a lane is a little state machine -- per loop iteration it advances on its chain
// word as far as registers allow, then issues at most ONE memory request group (root-table entry | node word 0 | node words 1-3 |
// two spill slots | chain word), all lanes wait ONCE, and each interprets what it asked for; (b) a wave owns a contiguous range of
// cursors and a lane that finishes one takes the next (wave-uniform counter, ballot rank -- no atomics), so every lane stays busy
// until the range is used up; (c) a cursor's tokens are read 16 at a time as ONE 64-byte line (stream-major [B][T], what a caller
// that holds B token sequences has anyway), requested one window ahead and parked in LDS, so no lane ever waits for tokens.
// The transitions, the visited-state count and the chain-word fetch rule are st_transfer_chain's, statement for statement.
// ================================================================================================
enum { N_NONE = 0, N_ROOT, N_NODE, N_TAIL, N_SPILL, N_CHAIN };
#define WALK_WIN 16            // tokens per window = one 64-byte line of a stream-major token row

template <int W, bool TRACE>
__global__ __launch_bounds__(256) void k_static_walk_streams(StaticDev S, int32_t *__restrict__ cursors, const int32_t *__restrict__ tokens,
                                                             int B, int T, int commit, int32_t *__restrict__ trace,
                                                             unsigned long long *__restrict__ visited_total, int per_wave) {
    __shared__ int tokwin[4][WALK_WIN][64];                  // [wave][slot][lane]: any slots of 64 lanes = 64 different 4-byte columns
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long wg = (long long)blockIdx.x * 4 + wave;
    long long next = wg * per_wave;                          // wave-uniform: first cursor nobody has taken yet
    long long c_end = next + per_wave; c_end = c_end < B ? c_end : B;
    int (*win)[64] = tokwin[wave];
    const int32_t *tok_last = tokens + ((size_t)B * T - WALK_WIN);      // last address a 64-byte window load may start at (the launcher checks B * T >= 16)
    const int root_last = S.vocab - 4;                       // (the launcher sends vocabularies < 4 to the lock-step kernel)

    int b = -1, t = 0, wend = 0, idx = 0, len = 0, tok = 0, need = N_NONE, lnk = 0;
    bool hopped = false;
    ChainWord cw = chain_none();
    int sp_base = 0, sp_h = 0, sp_m = 4, sp_probes = 0, root_at = 0;
    // the window requested ahead: cursor nb, tokens [nt0, nt0 + 16) of it, and its stored cursor
    bool have_next = false, want_prefetch = true;
    int nb = -1, nt0 = 0, nshift = 0;
    uint4 p0 = make_uint4(0, 0, 0, 0), p1 = p0, p2 = p0, p3 = p0;
    int2 pc = make_int2(0, 0);
    unsigned long long visited = 0;
    const unsigned term = W == 8 ? 0xFFFFu : 0xFFFFFFFFu;

    auto emit = [&]() { if (TRACE) reinterpret_cast<int2 *>(trace)[(size_t)b * T + t] = make_int2(idx, len); t++; };

    for (;;) {
        // ---- A1. window switch (at most one per iteration) ------------------------------------------------------------------
        if (need == N_NONE && (b < 0 || t == wend)) {
            const bool cursor_done = b >= 0 && t == T;
            if (cursor_done && commit) reinterpret_cast<int2 *>(cursors)[b] = make_int2(idx, len);
            if (have_next && nb >= 0) {
                // slot j holds token nt0 + j; a window loaded `nshift` tokens early (end of the matrix) has it at position j + nshift
                const int sh = nshift;
                win[(0 - sh) & 15][lane] = (int)p0.x; win[(1 - sh) & 15][lane] = (int)p0.y; win[(2 - sh) & 15][lane] = (int)p0.z; win[(3 - sh) & 15][lane] = (int)p0.w;
                win[(4 - sh) & 15][lane] = (int)p1.x; win[(5 - sh) & 15][lane] = (int)p1.y; win[(6 - sh) & 15][lane] = (int)p1.z; win[(7 - sh) & 15][lane] = (int)p1.w;
                win[(8 - sh) & 15][lane] = (int)p2.x; win[(9 - sh) & 15][lane] = (int)p2.y; win[(10 - sh) & 15][lane] = (int)p2.z; win[(11 - sh) & 15][lane] = (int)p2.w;
                win[(12 - sh) & 15][lane] = (int)p3.x; win[(13 - sh) & 15][lane] = (int)p3.y; win[(14 - sh) & 15][lane] = (int)p3.z; win[(15 - sh) & 15][lane] = (int)p3.w;
                if (nt0 == 0) { idx = pc.x; len = pc.y; cw = chain_none(); }
                b = nb; t = nt0; wend = nt0 + WALK_WIN < T ? nt0 + WALK_WIN : T;
                have_next = false; want_prefetch = true;
            } else if (have_next || b < 0 || cursor_done) { b = -1; have_next = false; }     // range used up / nothing requested yet
        }
        // ---- A2. advance without memory: negative tokens and chain-word transitions, up to the window's end -------------------
#pragma unroll 1
        for (int s = 0; s <= W; s++) {
            if (need != N_NONE || b < 0 || t == wend) break;
            tok = win[t & (WALK_WIN - 1)][lane];
            if (tok < 0) { idx = 0; len = 0; cw = chain_none(); visited++; emit(); continue; }    // no state has an edge on a negative token
            const unsigned nx = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
            if (nx != term && nx == (unsigned)tok) {                                             // register path of st_transfer_chain
                idx += 1; len += 1; visited++;
                if (W == 8) { cw.lo = (cw.lo >> 16) | (cw.hi << 48); cw.hi = (cw.hi >> 16) | (0xFFFFull << 48); }
                else { cw.lo = (cw.lo >> 32) | (cw.hi << 32); cw.hi = (cw.hi >> 32) | (0xFFFFFFFFull << 32); }
                emit();
                if (++cw.used == W) need = N_CHAIN;
                continue;
            }
            cw = chain_none(); hopped = false;
            need = idx == 0 ? N_ROOT : N_NODE;
        }
        if (__ballot(b >= 0 || have_next || want_prefetch) == 0ull) break;       // every lane idle, nothing in flight

        // ---- B. issue: every lane one 16-byte request (idle lanes re-read a line they hold), tails and windows on top -----------
        // Branch-free on purpose: loads inside divergent branches make the compiler's wait-count pass put a full wait in front of
        // each of them (first version of this kernel: 11 vmcnt(0) per iteration, 0.73 ms per launch).
        const uint4 *a0 = reinterpret_cast<const uint4 *>(S.nodes + idx);
        if (need == N_ROOT) {
            int g = (tok < S.vocab ? tok : 0) & ~3; g = g < root_last ? g : root_last;
            root_at = g; a0 = reinterpret_cast<const uint4 *>(S.root_next + g);
        }
        a0 = need == N_TAIL ? a0 + 1 : a0;
        // two adjacent slots of the hashed block per request; at the block's last slot the pair is (last - 1, last): never past the end
        a0 = need == N_SPILL ? reinterpret_cast<const uint4 *>(S.spill + sp_base + SAMD_SPILL_HEAD + (sp_h + 1 < sp_m ? sp_h : sp_h - 1)) : a0;
        a0 = need == N_CHAIN ? S.chain + idx : a0;
        const uint4 ld0 = *a0;
        uint4 ld1 = make_uint4(0, 0, 0, 0), ld2 = ld1;
        if (__ballot(need == N_TAIL) != 0ull) {               // wave-uniform
            const uint4 *a1 = need == N_TAIL ? a0 + 1 : a0, *a2 = need == N_TAIL ? a0 + 2 : a0;
            ld1 = *a1; ld2 = *a2;
        }
        {
            // next window of the same cursor, or the first window of a new one (rank among the lanes that ask now)
            const bool same = want_prefetch && b >= 0 && wend < T;
            const bool fresh = want_prefetch && !same;
            const unsigned long long fm = __ballot(fresh);
            if (__ballot(want_prefetch) != 0ull) {            // wave-uniform
                int qb = nb, qt = nt0;
                if (same) { qb = b; qt = wend; }
                else if (fresh) {
                    const long long mine = next + __popcll(fm & ((1ull << lane) - 1ull));
                    qb = mine < c_end ? (int)mine : -1; qt = 0;
                }
                const int rb = qb >= 0 ? qb : 0;
                const int32_t *row = tokens + (size_t)rb * T + qt;
                // a window that would run past the end of the token matrix (the last cursor's last window) is read from the last 64
                // bytes of the matrix instead and rotated into place when it is written to LDS (`shift` slots)
                const int32_t *r0 = row <= tok_last ? row : tok_last;
                if (want_prefetch) {
                    nshift = (int)(row - r0);
                    if ((T & 3) == 0) {
                        p0 = *reinterpret_cast<const uint4 *>(r0); p1 = *reinterpret_cast<const uint4 *>(r0 + 4);
                        p2 = *reinterpret_cast<const uint4 *>(r0 + 8); p3 = *reinterpret_cast<const uint4 *>(r0 + 12);
                    } else {                                  // rows that are not 16-byte aligned: 4-byte loads
                        p0 = make_uint4(r0[0], r0[1], r0[2], r0[3]); p1 = make_uint4(r0[4], r0[5], r0[6], r0[7]);
                        p2 = make_uint4(r0[8], r0[9], r0[10], r0[11]); p3 = make_uint4(r0[12], r0[13], r0[14], r0[15]);
                    }
                    pc = reinterpret_cast<const int2 *>(cursors)[rb];
                    nb = qb; nt0 = qt; have_next = true; want_prefetch = false;
                }
            }
            next += __popcll(fm);
        }

        // ---- C. interpret (one wait for everything issued above) ----------------------------------------------------------------
        if (need == N_CHAIN) {
            cw.lo = (unsigned long long)ld0.x | ((unsigned long long)ld0.y << 32);
            cw.hi = (unsigned long long)ld0.z | ((unsigned long long)ld0.w << 32);
            cw.used = 0; need = N_NONE;
        } else if (need != N_NONE) {
            int found = -1;                      // >= 0: edge target; -2: no edge here, follow the suffix link
            bool chain_after = false;
            if (need == N_ROOT) {
                visited++;
                const int k = tok - root_at;
                int nx = k == 0 ? (int)ld0.x : (k == 1 ? (int)ld0.y : (k == 2 ? (int)ld0.z : (int)ld0.w));
                nx = tok < S.vocab ? nx : -1;
                if (nx >= 0) { idx = nx; len += 1; } else { idx = 0; len = 0; }
                emit(); need = N_NONE;            // (no chain word after a landing through the root table: st_transfer_chain)
            } else if (need == N_NODE) {
                visited++;
                if (hopped) len = (int)ld0.y & SAMD_LEN_MASK;
                lnk = (int)ld0.x;
                if ((int)ld0.z == tok) { found = (int)ld0.w; chain_after = ((int)ld0.y & SAMD_RUN) != 0; }
                else if (!((int)ld0.y & SAMD_SINGLE)) need = N_TAIL;
                else found = -2;
            } else if (need == N_TAIL) {
                int nx = -1;
                nx = ((int)ld0.z == tok) ? (int)ld0.w : nx;       // w1 = {aux, deg, e1}
                nx = ((int)ld1.x == tok) ? (int)ld1.y : nx;       // w2 = {e2, e3}
                nx = ((int)ld1.z == tok) ? (int)ld1.w : nx;
                nx = ((int)ld2.x == tok) ? (int)ld2.y : nx;       // w3 = {e4, spill, reserved}
                if (nx < 0 && (int)ld0.y > SAMD_INLINE_EDGES) {
                    sp_base = (int)ld2.z; sp_m = (int)samd_spill_slots((int)ld0.y); sp_h = (int)samd_spill_hash(tok, (uint32_t)sp_m); sp_probes = 0;
                    need = N_SPILL;
                } else { found = nx >= 0 ? nx : -2; chain_after = true; }
            } else {                                              // N_SPILL: slots sp_h and sp_h + 1 of the open-addressing block
                const bool second = sp_h + 1 < sp_m;
                const int t0 = second ? (int)ld0.x : (int)ld0.z, d0 = second ? (int)ld0.y : (int)ld0.w, t1 = (int)ld0.z, d1 = (int)ld0.w;
                if (t0 == tok) found = d0;
                else if (t0 == -1) found = -2;
                else if (second && t1 == tok) found = d1;
                else if (second && t1 == -1) found = -2;
                else {
                    sp_probes += second ? 2 : 1;
                    sp_h = (sp_h + (second ? 2 : 1)) & (sp_m - 1);
                    if (sp_probes >= sp_m) found = -2;
                }
                chain_after = true;
            }
            if (found >= 0) {                                     // "edge found: follow, length + 1"
                idx = found; len += 1; emit();
                need = (chain_after && idx > 0) ? N_CHAIN : N_NONE;
            } else if (found == -2) {                             // "state <- link, length <- states[link].length"
                idx = lnk; hopped = true;
                if (idx == 0) len = 0;
                need = idx == 0 ? N_ROOT : N_NODE;
            }
        }
    }
    if (visited_total) {
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if (lane == 0 && visited) atomicAdd(visited_total, visited);
    }
}

// chain words from the node image: one thread per state (samd_common.h, CHAIN WORDS)
template <int W>
__global__ __launch_bounds__(256) void k_build_chain(const SamNode *__restrict__ nodes, long long n, uint4 *__restrict__ chain) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    unsigned tokw[W];
    bool alive = true;
#pragma unroll
    for (int j = 0; j < W; j++) {
        tokw[j] = W == 8 ? 0xFFFFu : 0xFFFFFFFFu;
        if (alive && s + j < n) {
            const int4 w0 = reinterpret_cast<const int4 *>(nodes + s + j)[0];
            if (w0.z >= 0 && (long long)w0.w == s + j + 1) tokw[j] = (unsigned)w0.z; else alive = false;
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
       OP_BUFFERS_ONLY = 4096 };

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
    int index, match, start, source, type, reverse;
    int c0, c1, c2, c3;         // OP_SET_CURSORS
    int have_static;
    int only_if_deferred;       // OP_SET_DRAFT: install only when the last lookup returned type 2
};

__device__ __forceinline__ void load_draft(const SessionDev &D, StepShared &sh, int &type, int &n, int &nl, int &md) {
    const int i = lane_id();
    type = D.dmeta[D_TYPE]; n = D.dmeta[D_N]; nl = D.dmeta[D_NLEAVES]; md = D.dmeta[D_MAXDEPTH];
    if (i < n) { sh.tokens[i] = D.tokens[i]; sh.parent[i] = D.parent[i]; }
    for (int k = i; k < SAMD_MAX_DRAFT * SAMD_MAX_DRAFT; k += WAVE) (&sh.path[0][0])[k] = PATH_PAD;
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

    if (A.ops & OP_ACCEPT) {
        int type, n, nl, md, a, nt;
        load_draft(D, sh, type, n, nl, md);
        do_accept(D, sh, A.node_argmax, type, n, nl, md, a, nt);
        wave_mem_sync();
    }
    if (A.ops & OP_COMMIT) {
        // DraftModel.update(accepted tokens) (draft.py:62-67)
        const int a = D.verdict[V_ACCEPT];
        if (lane < a) sh.accepted[lane] = D.acc_tokens[lane];
        __syncthreads();
        dyn_add_tokens(D, sh.accepted, a);
        if (A.have_static) {
            int is = D.meta[M_ST_IDX], ms = D.meta[M_ST_LEN];
            for (int i = 0; i < a; i++) st_transfer(S, is, ms, sh.accepted[i]);
            if (lane == 0) { D.meta[M_ST_IDX] = is; D.meta[M_ST_LEN] = ms; }
        }
        wave_mem_sync();
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
        if (A.have_static) for (int i = 0; i < n_in; i++) st_transfer(S, is, ms, A.tokens[i]);
        else if (n_in > 0) { is = 0; ms = 0; }
        if (A.commit && lane == 0) { D.meta[M_ST_IDX] = is; D.meta[M_ST_LEN] = ms; }
        if (A.out2 && lane == 0) { A.out2[0] = is; A.out2[1] = ms; }
        wave_mem_sync();
    }
    if (A.ops & OP_DRAFT) {
        do_draft(D, S, A.have_static != 0, P, sh, A.start_token[0]);
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
        if (lane < A.n) { sh.tokens[lane] = A.tokens[lane]; sh.parent[lane] = A.parents[lane]; }
        __syncthreads();
        int nl, mxd; build_buffers(sh, A.n, A.reverse, nl, mxd);
        store_draft(D, sh, A.type, A.n, nl, mxd, 0, 0, 0, 0, A.reverse);
    }
}

// standalone gen_buffers on a parent array
__global__ __launch_bounds__(64) void k_tree_buffers(const int32_t *parent, int n, int reverse, int32_t *position, uint64_t *mask,
                                                     uint8_t *mask_bool, int32_t *retrieve, int32_t *shape) {
    __shared__ StepShared sh;
    const int i = lane_id();
    if (i < n) sh.parent[i] = parent[i];
    __syncthreads();
    int nl, md; build_buffers(sh, n, reverse, nl, md);
    if (i < n) {
        if (position) position[i] = sh.position[i];
        if (mask) mask[i] = sh.mask[i];
        if (mask_bool) for (int j = 0; j < n; j++) mask_bool[(size_t)i * n + j] = (uint8_t)((sh.mask[i] >> j) & 1ull);
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
    if (i < n) { tok[i] = D.tokens[i]; par[i] = D.parent[i]; }
    __syncthreads();
    // is the committed history a prefix of target?
    int bad = 0;
    for (int k = i; k < nc; k += WAVE) bad |= (k >= n_target || hist[k] != target[k]);
    const bool hist_ok = __ballot(bad != 0) == 0ull;
    int res = 0;
    if (i < n) {
        int path[SAMD_MAX_DRAFT]; int d = 0;
        for (int j = i; j != -1; j = par[j]) path[d++] = tok[j];   // leaf -> root
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
    out[i] = res;
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
static void launch_walk(const samd_static_t *sam, int blocks, int threads, hipStream_t st, int32_t *d_cursors, const int32_t *d_tokens, int B, int T,
                        int commit, int32_t *d_trace, unsigned long long *d_visited) {
    static const bool use_chain = [] { const char *e = getenv("SAMD_WALK_CHAIN"); return !(e && e[0] == '0'); }();
    const StaticDev v = static_view(sam);
    if (!use_chain || !v.chain) hipLaunchKernelGGL((k_static_walk<8, false>), dim3(blocks), dim3(threads), 0, st, v, d_cursors, d_tokens, B, T, commit, d_trace, d_visited);
    else if (v.chain_w == 8) hipLaunchKernelGGL((k_static_walk<8, true>), dim3(blocks), dim3(threads), 0, st, v, d_cursors, d_tokens, B, T, commit, d_trace, d_visited);
    else hipLaunchKernelGGL((k_static_walk<4, true>), dim3(blocks), dim3(threads), 0, st, v, d_cursors, d_tokens, B, T, commit, d_trace, d_visited);
}

// root-child hash (samd_common.h): sizes, then fill.  One thread per vocabulary id.
__global__ __launch_bounds__(256) void k_d1_sizes(const SamNode *__restrict__ nodes, const int32_t *__restrict__ root_next, int vocab, int32_t *__restrict__ sizes) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= vocab) return;
    const int dst = root_next[tok];
    int m = 0;
    if (dst > 0) {
        const int deg = nodes[dst].deg;
        if (deg > SAMD_INLINE_EDGES) { m = 8; while (m < 2 * deg) m <<= 1; }
    }
    sizes[tok] = m;
}
__global__ __launch_bounds__(256) void k_d1_fill(const SamNode *__restrict__ nodes, const SamEdge *__restrict__ spill, const int32_t *__restrict__ root_next,
                                                 int vocab, const long long *__restrict__ offsets, const int32_t *__restrict__ sizes,
                                                 uint4 *__restrict__ root16, SamEdge *__restrict__ hash) {
    const int tok = blockIdx.x * blockDim.x + threadIdx.x;
    if (tok >= vocab) return;
    const int dst = root_next[tok], m = sizes[tok];
    const long long base = offsets[tok];
    root16[tok] = make_uint4((unsigned)dst, (unsigned)base, (unsigned)m, 0u);
    if (m == 0) return;
    SamEdge *tab = hash + base;
    auto put = [&](int t, int d) {
        if (t < 0) return;
        uint32_t h = samd_spill_hash(t, (uint32_t)m);
        while (tab[h].tok != -1) { if (tab[h].tok == t) return; h = (h + 1) & (uint32_t)(m - 1); }     // (the spill head repeats ranks 5..7)
        tab[h].tok = t; tab[h].dst = d;
    };
    const int *w = reinterpret_cast<const int *>(nodes + dst);
    for (int k = 0; k < SAMD_INLINE_EDGES; k++) put(w[SAMD_EDGE_WORD(k)], w[SAMD_EDGE_WORD(k) + 1]);
    const int deg = w[5];
    const SamEdge *sp = spill + w[14];
    const uint32_t slots = samd_spill_slots(deg);
    for (uint32_t k = 0; k < SAMD_SPILL_HEAD + slots; k++) put(sp[k].tok, sp[k].dst);
}

static int derive_root_hash(samd_static_t *s, hipStream_t st) {
    static const bool enabled = [] { const char *e = getenv("SAMD_ROOT_HASH"); return !(e && e[0] == '0'); }();      // A/B switch, read once
    if (!enabled || s->vocab < 1 || s->vocab > (1 << 24)) return SAMD_OK;
    if (s->d_root16) { (void)hipFree(s->d_root16); s->d_root16 = nullptr; }
    if (s->d_d1hash) { (void)hipFree(s->d_d1hash); s->d_d1hash = nullptr; }
    const int vocab = (int)s->vocab;
    int32_t *d_sizes = nullptr; long long *d_off = nullptr;
    if (hipMalloc((void **)&d_sizes, (size_t)vocab * 4) != hipSuccess || hipMalloc((void **)&d_off, (size_t)vocab * 8) != hipSuccess) {
        if (d_sizes) (void)hipFree(d_sizes);
        samd_set_error("hipMalloc(root-child hash sizes) failed"); return SAMD_E_HIP;
    }
    const unsigned blocks = (unsigned)((vocab + 255) / 256);
    hipLaunchKernelGGL(k_d1_sizes, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_root, vocab, d_sizes);
    std::vector<int32_t> sizes(vocab);
    std::vector<long long> off(vocab);
    int rc = SAMD_OK;
    if (hipMemcpyAsync(sizes.data(), d_sizes, (size_t)vocab * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
    long long total = 0;
    for (int i = 0; i < vocab && rc == SAMD_OK; i++) { off[i] = total; total += sizes[i]; }
    if (rc == SAMD_OK && total >= (1ll << 31)) { total = 0; rc = -1; }      // base does not fit 32 bits: go without the hash
    if (rc == SAMD_OK && total > 0) {
        if (hipMalloc(&s->d_root16, (size_t)vocab * 16) != hipSuccess || hipMalloc(&s->d_d1hash, (size_t)total * 8) != hipSuccess) rc = SAMD_E_HIP;
        if (rc == SAMD_OK && (hipMemsetAsync(s->d_d1hash, 0xFF, (size_t)total * 8, st) != hipSuccess ||
                              hipMemcpyAsync(d_off, off.data(), (size_t)vocab * 8, hipMemcpyHostToDevice, st) != hipSuccess)) rc = SAMD_E_HIP;
        if (rc == SAMD_OK) {
            hipLaunchKernelGGL(k_d1_fill, dim3(blocks), dim3(256), 0, st, s->d_nodes, s->d_spill, s->d_root, vocab, d_off, d_sizes, (uint4 *)s->d_root16, (SamEdge *)s->d_d1hash);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = SAMD_E_HIP;
        }
        s->n_d1hash = total;
    }
    (void)hipFree(d_sizes); (void)hipFree(d_off);
    if (rc != SAMD_OK) {
        if (s->d_root16) { (void)hipFree(s->d_root16); s->d_root16 = nullptr; }
        if (s->d_d1hash) { (void)hipFree(s->d_d1hash); s->d_d1hash = nullptr; }
        s->n_d1hash = 0;
        if (rc == SAMD_E_HIP) { samd_set_error("root-child hash derivation failed"); return rc; }
    }
    return SAMD_OK;
}

int samd_static_derive_chain(samd_static_t *s, void *stream) {
    if (!s || !s->uploaded || !s->d_nodes) return SAMD_E_INVALID;
    if (!s->d_chain && hipMalloc(&s->d_chain, (size_t)s->n_states * 16) != hipSuccess) { s->d_chain = nullptr; samd_set_error("hipMalloc(chain words) failed"); return SAMD_E_HIP; }
    const unsigned blocks = (unsigned)((s->n_states + 255) / 256);
    if (s->vocab < 65535) hipLaunchKernelGGL(k_build_chain<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->d_nodes, (long long)s->n_states, (uint4 *)s->d_chain);
    else hipLaunchKernelGGL(k_build_chain<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->d_nodes, (long long)s->n_states, (uint4 *)s->d_chain);
    LAUNCHCHK();
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { samd_set_error("chain-word derivation failed"); return SAMD_E_HIP; }
    return derive_root_hash(s, (hipStream_t)stream);
}

int samd_static_walk(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                     int32_t commit, int32_t *d_trace, void *stream) {
    if (!sam || !sam->uploaded || B < 0 || T < 0) { samd_set_error("samd_static_walk: invalid argument"); return SAMD_E_INVALID; }
    if (B == 0 || T == 0) return SAMD_OK;                 // empty batch / no tokens: nothing to do
    if (!d_cursors || !d_tokens) { samd_set_error("samd_static_walk: null pointer"); return SAMD_E_INVALID; }
    const int threads = 256, blocks = (B + threads - 1) / threads;
    launch_walk(sam, blocks, threads, (hipStream_t)stream, d_cursors, d_tokens, B, T, commit, d_trace, (unsigned long long *)nullptr);
    LAUNCHCHK();
    return SAMD_OK;
}

// same launch, additionally accumulating the number of visited states into *d_visited (u64, device);
// used by bench.py to turn kernel time into algorithmic bytes (16 B per visited state).
int samd_static_walk_counted(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t commit, uint64_t *d_visited, void *stream) {
    if (!sam || !sam->uploaded || !d_cursors || !d_tokens || B <= 0 || T <= 0 || !d_visited) return SAMD_E_INVALID;
    const int threads = 256, blocks = (B + threads - 1) / threads;
    launch_walk(sam, blocks, threads, (hipStream_t)stream, d_cursors, d_tokens, B, T, commit, (int32_t *)nullptr, (unsigned long long *)d_visited);
    LAUNCHCHK();
    return SAMD_OK;
}

// fewer than 16 tokens in all, or a vocabulary below 4 ids: the lock-step kernel on a time-major copy (B * T < 16 elements, or a toy automaton)
__global__ void k_transpose_small(const int32_t *src, int32_t *dst, int rows, int cols, int width) {      // dst[c][r] = src[r][c], `width` ints per element
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols;
    for (int k = 0; k < width; k++) dst[((size_t)c * rows + r) * width + k] = src[((size_t)r * cols + c) * width + k];
}
static int walk_streams_small(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T, int32_t commit,
                              int32_t *d_trace, uint64_t *d_visited, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int32_t *tm = nullptr, *tr = nullptr;
    const size_t n = (size_t)B * T;
    if (hipMallocAsync((void **)&tm, n * 4, st) != hipSuccess) { samd_set_error("samd_static_walk_streams: scratch allocation failed"); return SAMD_E_HIP; }
    if (d_trace && hipMallocAsync((void **)&tr, n * 8, st) != hipSuccess) { (void)hipFreeAsync(tm, st); samd_set_error("samd_static_walk_streams: scratch allocation failed"); return SAMD_E_HIP; }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_transpose_small, dim3(blocks), dim3(256), 0, st, d_tokens, tm, B, T, 1);
    launch_walk(sam, (B + 255) / 256, 256, st, d_cursors, tm, B, T, commit, tr, (unsigned long long *)d_visited);
    if (d_trace) hipLaunchKernelGGL(k_transpose_small, dim3(blocks), dim3(256), 0, st, tr, d_trace, T, B, 2);
    (void)hipFreeAsync(tm, st);
    if (tr) (void)hipFreeAsync(tr, st);
    LAUNCHCHK();
    return SAMD_OK;
}

// stream-major form: d_tokens int32 [B][T], d_trace int32 [B][T][2]; lanes decoupled, cursors handed out dynamically inside a wave
static int walk_streams(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T, int32_t commit,
                        int32_t *d_trace, uint64_t *d_visited, void *stream) {
    if (!sam || !sam->uploaded || B < 0 || T < 0) { samd_set_error("samd_static_walk_streams: invalid argument"); return SAMD_E_INVALID; }
    if (B == 0 || T == 0) return SAMD_OK;
    if (!d_cursors || !d_tokens) { samd_set_error("samd_static_walk_streams: null pointer"); return SAMD_E_INVALID; }
    if (!sam->d_chain) { samd_set_error("samd_static_walk_streams: chain words missing (upload / adopt derives them)"); return SAMD_E_INVALID; }
    if (sam->vocab < 4 || (long long)B * T < 16) return walk_streams_small(sam, d_cursors, d_tokens, B, T, commit, d_trace, d_visited, stream);
    // cursors per wave: enough waves to fill the chip a few times over (256 CUs x 32 wave slots), each with several cursors per lane
    // so that the dynamic hand-out has something to balance; SAMD_WALK_PER_WAVE overrides (read once)
    static const int env_pw = [] { const char *e = getenv("SAMD_WALK_PER_WAVE"); return e ? atoi(e) : 0; }();
    int per_wave = env_pw > 0 ? env_pw : 256;
    while (per_wave > 64 && env_pw <= 0 && (long long)(B + per_wave - 1) / per_wave < 4096) per_wave >>= 1;
    const long long waves = ((long long)B + per_wave - 1) / per_wave;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const StaticDev v = static_view(sam);
#define GO(WW, TR) hipLaunchKernelGGL((k_static_walk_streams<WW, TR>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, d_cursors, d_tokens, B, T, commit, d_trace, \
                                      (unsigned long long *)d_visited, per_wave)
    if (v.chain_w == 8) { if (d_trace) GO(8, true); else GO(8, false); }
    else { if (d_trace) GO(4, true); else GO(4, false); }
#undef GO
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
                 o_tok = carve(4 * 64), o_par = carve(4 * 64), o_pos = carve(4 * 64), o_mask = carve(8 * 64), o_ret = carve(4 * 64 * 64),
                 o_rep = carve(4 * SAMD_REPORT_INTS), o_st = carve(4), o_cl = carve(4);
    s->arena_bytes = off;
    if (hipMalloc(&s->arena, off) != hipSuccess) { free(s); samd_set_error("hipMalloc(session arena) failed"); return SAMD_E_HIP; }
    char *base = (char *)s->arena;
    D.link = (int32_t *)(base + o_link); D.length = (int32_t *)(base + o_len); D.minend = (int32_t *)(base + o_me);
    D.head = (int32_t *)(base + o_head); D.tail = (int32_t *)(base + o_tail); D.hkey = (uint64_t *)(base + o_hk);
    D.hdst = (int32_t *)(base + o_hd); D.hnext = (int32_t *)(base + o_hn); D.text = (int32_t *)(base + o_text);
    D.tokens = (int32_t *)(base + o_tok); D.parent = (int32_t *)(base + o_par);
    D.position = (int32_t *)(base + o_pos); D.mask = (uint64_t *)(base + o_mask); D.retrieve = (int32_t *)(base + o_ret);
    int32_t *rep = (int32_t *)(base + o_rep);
    D.dmeta = rep + SAMD_REP_DMETA; D.verdict = rep + SAMD_REP_VERDICT; D.acc_tokens = rep + SAMD_REP_TOKENS;
    D.kv_index = rep + SAMD_REP_KVINDEX; D.counters = rep + SAMD_REP_COUNTERS; D.meta = rep + SAMD_REP_META;
    D.start_token = (int32_t *)(base + o_st); D.cache_length = (int32_t *)(base + o_cl);
    if (hipMemset(s->arena, 0, off) != hipSuccess) { (void)hipFree(s->arena); free(s); return SAMD_E_HIP; }
    *out = s;
    int rc = samd_session_reset(s, nullptr);
    if (rc == SAMD_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = SAMD_E_HIP;
    if (rc) { samd_session_free(s); *out = nullptr; }
    return rc;
}

void samd_session_free(samd_session_t *s) {
    if (!s) return;
    if (s->arena) (void)hipFree(s->arena);
    free(s);
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
    return launch_session(s, sam, p, A, stream);
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
    D2H(out->tokens, D.tokens, 4 * 64); D2H(out->parent, D.parent, 4 * 64); D2H(out->position, D.position, 4 * 64);
    D2H(out->mask, D.mask, 8 * 64); D2H(out->retrieve, D.retrieve, 4 * 64 * 64);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    out->type = dm[D_TYPE]; out->n = dm[D_N]; out->n_leaves = dm[D_NLEAVES]; out->max_depth = dm[D_MAXDEPTH];
    out->index_dyn = dm[D_IDX_DYN]; out->match_dyn = dm[D_MATCH_DYN]; out->index_static = dm[D_IDX_ST]; out->match_static = dm[D_MATCH_ST];
    return SAMD_OK;
}

int samd_session_read_verdict(samd_session_t *s, samd_verdict_host_t *out, void *stream) {
    if (!s || !out) return SAMD_E_INVALID;
    const SessionDev &D = s->dev;
    int32_t v[V_COUNT];
    D2H(v, D.verdict, sizeof(v)); D2H(out->tokens, D.acc_tokens, 4 * 64); D2H(out->kv_index, D.kv_index, 4 * 64);
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

int samd_tree_buffers(const int32_t *d_parent, int32_t n, int32_t reverse_leaves, int32_t *d_position, uint64_t *d_mask,
                      uint8_t *d_mask_bool, int32_t *d_retrieve, int32_t *d_shape, void *stream) {
    if (!d_parent || n < 1 || n > SAMD_MAX_DRAFT) { samd_set_error("samd_tree_buffers: n must be in [1,64]"); return SAMD_E_INVALID; }
    hipLaunchKernelGGL(k_tree_buffers, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, d_parent, n, reverse_leaves, d_position, d_mask,
                       d_mask_bool, d_retrieve, d_shape);
    LAUNCHCHK();
    return SAMD_OK;
}

}  // extern "C"
