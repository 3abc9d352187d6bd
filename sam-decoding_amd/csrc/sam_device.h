// sam_device.h -- gfx950 device functions of the suffix-automaton path (included by sam_kernels.hip).
//
// Two execution shapes share this code:
//   * lane-per-stream (batched walk): every lane owns an independent cursor; st_transfer() is ordinary
//     per-lane code, each visited state costs one 64-byte line (4 x dwordx4 from the same line).
//   * single wavefront per request (bs=1 decode step): the 64 lanes run the sequential control flow
//     uniformly (uniform addresses -> one request per load) and fan out where the work is parallel:
//     64 hash slots per probe of the dynamic automaton (ballot), 8 successors per tree expansion,
//     one tree node / leaf row per lane for the buffers and the greedy posterior.
#pragma once
#include <hip/hip_runtime.h>
#include "samd_common.h"

#define WAVE 64
#define PATH_PAD 255

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// make the wave's own earlier global stores visible to its later loads (single-CU, workgroup scope)
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ------------------------------------------------------------------------------------------------
// static automaton: one longest-suffix-match transition
// reference: transfer_state, samd_sam_only/sam/static_sam.py:98-107 (== samd/sam/static_sam.py:81-90)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int spill_search(const SamEdge *tab, int deg, int tok) {
    const uint32_t m = samd_spill_slots(deg);       // open addressing, load factor <= 1/2 (samd_common.h)
    uint32_t h = samd_spill_hash(tok, m);
    for (uint32_t probes = 0; probes < m; probes++) {
        const SamEdge e = tab[h];
        if (e.tok == tok) return e.dst;
        if (e.tok == -1) return -1;
        h = (h + 1) & (m - 1);
    }
    return -1;
}

// returns the number of states visited (for the bytes-per-visit accounting of the bench).
// Common case = ONE 16-byte load per visited state (w0: link, length, most frequent successor); the rest of the
// 64-byte node is read only when that successor is not the token and the state has more edges.
__device__ __forceinline__ int st_transfer(const StaticDev &S, int &idx, int &len, int tok) {
    int visited = 0;
    bool hopped = false;
    if (tok < 0) { idx = 0; len = 0; return 1; }   // no state has an edge on a negative token
    for (;;) {
        visited++;
        if (idx == 0) {                            // root: dense table; failing here ends in (0,0)
            int nx = (tok < S.vocab) ? S.root_next[tok] : -1;
            if (nx >= 0) { idx = nx; len += 1; } else { idx = 0; len = 0; }
            return visited;
        }
        const int4 *np = reinterpret_cast<const int4 *>(S.nodes + idx);
        const int4 w0 = np[0];
        if (hopped) len = w0.y & SAMD_LEN_MASK;    // length <- states[link].length (static_sam.py:101)
        int nx = (w0.z == tok) ? w0.w : -1;
        if (nx < 0 && !(w0.y & SAMD_SINGLE)) {
            const int4 w1 = np[1], w2 = np[2], w3 = np[3];
            nx = (w1.z == tok) ? w1.w : nx;
            nx = (w2.x == tok) ? w2.y : nx;
            nx = (w2.z == tok) ? w2.w : nx;
            nx = (w3.x == tok) ? w3.y : nx;
            if (nx < 0 && w1.y > SAMD_INLINE_EDGES)
                nx = spill_search(S.spill + w3.z + SAMD_SPILL_HEAD, w1.y, tok);
        }
        if (nx >= 0) { idx = nx; len += 1; return visited; }
        idx = w0.x; hopped = true;
        if (idx == 0) len = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// The same transition for a cursor that may hold a CHAIN WORD (samd_common.h): `cw` = the rank-0 tokens of the non-branching
// run that starts at the cursor's state, next one in the low bits, all-ones once exhausted.  A token that equals the word's
// next token IS the transition "edge found -> follow, length + 1" of transfer_state (the word repeats what the nodes say), so
// it costs no memory access; anything else drops the word and goes through the nodes exactly as st_transfer does.  After
// following a rank-0 edge whose source carries SAMD_RUN, after any edge of rank >= 1, and when a full word has been used up, the
// word of the new state is fetched (one 16-byte load that then serves up to W transitions).  W = 8 (u16 tokens) or 4 (u32).  Results are identical to st_transfer's by construction;
// tests/test_gpu_sam.py and test_gpu_fullsize.py compare both with the oracle.
// ------------------------------------------------------------------------------------------------
// Round 4: chain entries carry a FLAG in their top bit (clear = flagged, so that "no word" = all-ones says nothing): entry j of the word
// loaded at state s, a real token, flagged <=> state s + j has no edge but that token (SAMD_SINGLE) and its suffix link is a child of
// the root.  A cursor at s + j that holds the word and meets another token then knows the outcome of transfer_state's whole climb
// without loading anything at s + j: the state fails, its link is root_next[previous token] (the only root child whose strings end
// in the token that brought the cursor here), whose hashed block decides in one probe (root16 carries its length).  In the bench walk
// this is what every noise token does: node + link node + its tail + a spill probe (3.5 requests, 5 dependent rounds) become one
// probe.  W = 8: 15-bit tokens (vocabularies <= 32767); W = 4: 31-bit tokens.
// used = tokens of this word already consumed; (nlo, nhi) = the NEXT word of the run when have_next: requested while the last entry of
// this one is still unused, so that a cursor deep in a run never waits for a word (a wave otherwise stalls at the top of every token
// for the lanes -- one in eight -- that used up a word at the previous one)
// hub (round 5) = the cursor's state is BRANCHING and therefore in the edge table (samd_common.h): known from the table entry that led to it;
// its next transition probes the table without looking at the node
// ptok = the token of the transition that brought the cursor to its state (-1: unknown): what the flagged climb keys the bigram table with.  It
// lives IN the cursor's word so that it cannot go stale: st_transfer_chain sets it after every transition, a fresh cursor (chain_none) has none.
// hub carries, on the EDGE BLOCK path (round 6, samd_common.h), the cursor's state's BLOCK REFERENCE (never 0) instead of a plain 1
struct ChainWord { unsigned long long lo, hi, nlo, nhi; int used, have_next; unsigned hub; int ptok; };
__device__ __forceinline__ ChainWord chain_none() { ChainWord c; c.lo = c.hi = c.nlo = c.nhi = ~0ull; c.used = 0; c.have_next = 0; c.hub = 0; c.ptok = -1; return c; }
__device__ __forceinline__ ChainWord chain_load(const StaticDev &S, int state) {
    const uint4 c = S.chain[state];
    ChainWord w;
    w.lo = (unsigned long long)c.x | ((unsigned long long)c.y << 32);
    w.hi = (unsigned long long)c.z | ((unsigned long long)c.w << 32);
    w.nlo = w.nhi = ~0ull;
    w.used = 0; w.have_next = 0; w.hub = 0;
    return w;
}
// the first half of a state's chain word, as the root-child hash stores it beside the edge that leads there (samd_common.h): it serves
// W / 2 transitions; when those are used up the full word of the state reached is fetched, exactly as after a full word
template <int W>
__device__ __forceinline__ ChainWord chain_half(unsigned lo, unsigned hi) {
    ChainWord w;
    w.lo = (unsigned long long)lo | ((unsigned long long)hi << 32);
    w.hi = w.nlo = w.nhi = ~0ull;
    w.used = W / 2; w.have_next = 0; w.hub = 0;
    return w;
}
// the FIRST entry of a state's chain word (the bigram table's entries of large vocabularies have room for one): one transition, then the
// full word of the state reached is fetched
template <int W>
__device__ __forceinline__ ChainWord chain_first(unsigned first) {
    ChainWord w;
    w.lo = (unsigned long long)first | (W == 8 ? 0xFFFFFFFFFFFF0000ull : 0xFFFFFFFF00000000ull);
    w.hi = w.nlo = w.nhi = ~0ull;
    w.used = W - 1; w.have_next = 0; w.hub = 0;
    return w;
}
// what an EDGE TABLE entry carries of chain[dst]: its first 32 bits -- two entries (W = 8) or one (W = 4)
template <int W>
__device__ __forceinline__ ChainWord chain_from_edge(unsigned x) {
    if (W != 8) return chain_first<W>(x);
    ChainWord w;
    w.lo = (unsigned long long)x | 0xFFFFFFFF00000000ull;
    w.hi = w.nlo = w.nhi = ~0ull;
    w.used = W - 2; w.have_next = 0; w.hub = 0;
    return w;
}
// is (state, tok) in the edge table?  e = the slot already loaded at h; linear probing
__device__ __forceinline__ bool edge_find(const StaticDev &S, int state, int tok, uint32_t h, uint4 &e) {
    for (uint32_t probes = 0; probes <= S.edge_mask; probes++) {
        if (e.x == (unsigned)state && e.y == (unsigned)tok) return true;
        if (e.x == 0xFFFFFFFFu) return false;
        h = (h + 1) & S.edge_mask; e = S.ehash[h];
    }
    return false;
}

// A cursor that sits on the root child of token a is carried as idx = -2 - a (samd_common.h, BIGRAM TABLE) and resolved where an index is
// written out.
__device__ __forceinline__ bool st_on_child(int idx) { return idx <= -2; }
__device__ __forceinline__ int st_child_of(int tok) { return -2 - tok; }
__device__ __forceinline__ int st_resolve(const StaticDev &S, int idx) { return idx <= -2 ? (int)S.root16[-2 - idx].x : idx; }
// does token t have a root child?  `bits` = S.rc_bits or its copy in LDS
__device__ __forceinline__ bool st_has_child(const StaticDev &S, const uint32_t *bits, int t) {
    return t < S.vocab && ((bits[(unsigned)t >> 5] >> (t & 31)) & 1u);
}
// the root's transition on `tok`: no memory beyond the bitmap
__device__ __forceinline__ void st_from_root(const StaticDev &S, const uint32_t *bits, int tok, int &idx, int &len) {
    if (st_has_child(S, bits, tok)) { idx = st_child_of(tok); len = 1; } else { idx = 0; len = 0; }
}

// ptok = the token of the previous transition of this cursor (the one that brought it to idx), or -1 when unknown (cw.ptok, see ChainWord).
// `bits`: see st_has_child.  With the bigram table (S.bigram) the function may leave idx in the unresolved form (st_on_child).
template <int W>
__device__ __forceinline__ int st_transfer_chain_impl(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, int ptok, ChainWord &cw);
template <int W>
__device__ __forceinline__ int st_transfer_blocks_impl(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, int ptok, ChainWord &cw);
template <int W>
__device__ __forceinline__ int st_transfer_chain(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, ChainWord &cw) {
    const int visited = S.blocks ? st_transfer_blocks_impl<W>(S, bits, idx, len, tok, cw.ptok, cw)       // (uniform: one path per automaton)
                                 : st_transfer_chain_impl<W>(S, bits, idx, len, tok, cw.ptok, cw);
    cw.ptok = tok;                                           // (the word may have been replaced inside: set last)
    return visited;
}
// the block path alone, for a kernel that is instantiated for handles WITH blocks (k_static_walk<.., 2>): without the run-time choice the
// other path's code, its pointers (scalar registers: the kernel runs 8 waves per SIMD on 96 of them) and its branches are not compiled in
template <int W>
__device__ __forceinline__ int st_transfer_blocks(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, ChainWord &cw) {
    const int visited = st_transfer_blocks_impl<W>(S, bits, idx, len, tok, cw.ptok, cw);
    cw.ptok = tok;
    return visited;
}
template <int W>
__device__ __forceinline__ int st_transfer_chain_impl(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, int ptok, ChainWord &cw) {
    constexpr unsigned LOW = W == 8 ? 0x7FFFu : 0x7FFFFFFFu, HI = LOW + 1u;
    if (tok < 0) { idx = 0; len = 0; cw = chain_none(); return 1; }
    const unsigned ent = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
    const bool is_tok = (ent & LOW) != LOW;                  // a chain token (the end marker is all-ones)
    if (is_tok && (ent & LOW) == (unsigned)tok) {            // register path
        idx += 1; len += 1; cw.hub = 0;
        if (W == 8) { cw.lo = (cw.lo >> 16) | (cw.hi << 48); cw.hi = (cw.hi >> 16) | (0xFFFFull << 48); }
        else { cw.lo = (cw.lo >> 32) | (cw.hi << 32); cw.hi = (cw.hi >> 32) | (0xFFFFFFFFull << 32); }
        // a FULL word used up: the run may go on -- its next word was requested one token ago (below); without it (a word that came
        // in with one entry left) fetch the new state's word now.  A word that ended early marks the end of the run.
        if (++cw.used == W) {
            if (cw.have_next) { cw.lo = cw.nlo; cw.hi = cw.nhi; cw.nlo = cw.nhi = ~0ull; cw.used = 0; cw.have_next = 0; }
            else { cw = chain_load(S, idx); return 1; }
        }
        const unsigned e2 = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
        if (cw.used == W - 1 && (e2 & LOW) != LOW) {         // one entry left and the run goes on: the word of the state after it
            const uint4 c = S.chain[idx + 1];
            cw.nlo = (unsigned long long)c.x | ((unsigned long long)c.y << 32);
            cw.nhi = (unsigned long long)c.z | ((unsigned long long)c.w << 32);
            cw.have_next = 1;
        }
        return 1;
    }
    const bool have_table = S.bigram != nullptr;
    // flagged entry: idx has one edge, and it is not `tok`; its suffix link is the root child of ptok.  transfer_state visits idx, hops
    // (length <- states[link].length) and looks for `tok` there -- in the bigram table under (ptok, tok)
    const bool climbing = is_tok && !(ent & HI) && ptok >= 0 && have_table;
    const bool have_edges = S.ehash != nullptr;
    const bool at_hub = cw.hub && have_edges && idx > 0;
    cw = chain_none();
    int visited = climbing ? 1 : 0;
    const bool probing = climbing || st_on_child(idx);
    if (!probing && idx == 0 && have_table) {                // at the root: no request at all
        st_from_root(S, bits, tok, idx, len);
        return 1;
    }
    // THE FIRST LOAD OF EVERY LANE IS ONE INSTRUCTION: 16 bytes from a per-lane address -- a slot of the bigram table (the climbing lanes
    // and the lanes that sit on a root child: one probe decides) or word 0 of the cursor's node.  A wave advances in lock-step, so what it
    // pays per token is the number of dependent PHASES, not of loads: with the kinds in separate branches a token cost their sum
    // (profiles/r04_walk.md).
    const int a = climbing ? ptok : -2 - idx;
    uint32_t h = probing ? samd_bigram_hash(a, tok) & S.bigram_mask : (at_hub ? samd_edge_hash(idx, tok) & S.edge_mask : 0u);
    const uint4 *addr = probing ? S.bigram + h : (at_hub ? S.ehash + h : reinterpret_cast<const uint4 *>(S.nodes + idx));
    const uint4 first = *addr;
    if (probing) {
        uint4 e = first;
        bool hit = false;
        for (uint32_t probes = 0; probes <= S.bigram_mask; probes++) {
            if (W == 8) hit = (e.x & 0x3FFFFFFFu) == ((unsigned)a | ((unsigned)tok << 15)) && tok < 0x8000;
            else hit = (e.x & 0x7FFFFFFFu) == (unsigned)a && (e.y & 0x7FFFFFFFu) == (unsigned)tok;
            if (hit || e.x == 0xFFFFFFFFu) break;
            h = (h + 1) & S.bigram_mask; e = S.bigram[h];
        }
        if (hit) {
            if (climbing) {
                const unsigned lb = W == 8 ? e.x >> 30 : (e.x >> 31) | ((e.y >> 31) << 1);
                len = lb < 3 ? (int)lb + 1 : (int)S.root16[a].w;
            }
            len += 1;
            if (W == 8) { idx = (int)(e.y & 0x7FFFFFFFu); cw = chain_half<W>(e.z, e.w); cw.hub = (int)(e.y >> 31); }
            else { idx = (int)(e.z & 0x7FFFFFFFu); cw = chain_first<W>(e.w); cw.hub = (int)(e.z >> 31); }
            return visited + 1;
        }
        // no edge in the table = that state visited too, then the hop to ITS suffix link, the root
        st_from_root(S, bits, tok, idx, len);
        return visited + 2;
    }
    if (have_edges) {
        // ---- with the EDGE TABLE: a branching state answers in one probe; a climb issues a hop's node word 0 and its probe together ----------
        auto follow = [&](const uint4 &e) {
            idx = (int)(e.z & 0x7FFFFFFFu); len += 1;
            cw = chain_from_edge<W>(e.w); cw.hub = (int)(e.z >> 31);
        };
        auto from_root = [&]() {
            if (have_table) st_from_root(S, bits, tok, idx, len);
            else { const int nx = tok < S.vocab ? S.root_next[tok] : -1; if (nx >= 0) { idx = nx; len += 1; } else { idx = 0; len = 0; } }
        };
        visited++;
        if (idx == 0) { from_root(); return visited; }
        int link;
        if (at_hub) {                                         // the cursor's own state, known to branch: the probe decides; its node only on a miss
            uint4 e = first;
            if (edge_find(S, idx, tok, h, e)) { follow(e); return visited; }
            link = S.nodes[idx].link;
        } else {
            const int4 w0 = make_int4((int)first.x, (int)first.y, (int)first.z, (int)first.w);
            if (w0.z == tok) {                                // rank-0 edge
                idx = w0.w; len += 1;
                if (w0.y & SAMD_RUN) cw = chain_load(S, idx);
                return visited;
            }
            if (!(w0.y & SAMD_SINGLE)) {                      // more edges: all of them are in the table
                uint32_t h2 = samd_edge_hash(idx, tok) & S.edge_mask;
                uint4 e = S.ehash[h2];
                if (edge_find(S, idx, tok, h2, e)) { follow(e); return visited; }
            }
            link = w0.x;
        }
        idx = link;
        for (;;) {                                            // the climb (static_sam.py:99-101)
            visited++;
            if (idx == 0) { len = 0; from_root(); return visited; }
            const uint32_t h2 = samd_edge_hash(idx, tok) & S.edge_mask;
            const int4 w0 = reinterpret_cast<const int4 *>(S.nodes + idx)[0];       // two independent loads: one round per hop
            uint4 e = S.ehash[h2];
            len = w0.y & SAMD_LEN_MASK;                       // length <- states[link].length
            if (w0.y & SAMD_SINGLE) {
                if (w0.z == tok) {
                    idx = w0.w; len += 1;
                    if (w0.y & SAMD_RUN) cw = chain_load(S, idx);
                    return visited;
                }
            } else if (edge_find(S, idx, tok, h2, e)) { follow(e); return visited; }
            idx = w0.x;
        }
    }
    bool hopped = false, use_first = true;
    for (;;) {
        visited++;
        if (idx == 0) {
            if (have_table) st_from_root(S, bits, tok, idx, len);
            else {
                const int nx = tok < S.vocab ? S.root_next[tok] : -1;
                if (nx >= 0) { idx = nx; len += 1; } else { idx = 0; len = 0; }
            }
            return visited;
        }
        const int4 *np = reinterpret_cast<const int4 *>(S.nodes + idx);
        int4 w0;
        if (use_first) { w0 = make_int4((int)first.x, (int)first.y, (int)first.z, (int)first.w); use_first = false; }
        else w0 = np[0];
        if (hopped) len = w0.y & SAMD_LEN_MASK;
        if (w0.z == tok) {                                   // rank-0 edge
            idx = w0.w; len += 1;
            if (w0.y & SAMD_RUN) cw = chain_load(S, idx);
            return visited;
        }
        int nx = -1;
        if (!(w0.y & SAMD_SINGLE)) {
            const int4 w1 = np[1], w2 = np[2], w3 = np[3];
            nx = (w1.z == tok) ? w1.w : nx;
            nx = (w2.x == tok) ? w2.y : nx;
            nx = (w2.z == tok) ? w2.w : nx;
            nx = (w3.x == tok) ? w3.y : nx;
            if (nx < 0 && w1.y > SAMD_INLINE_EDGES)
                nx = spill_search(S.spill + w3.z + SAMD_SPILL_HEAD, w1.y, tok);
        }
        // an edge of rank >= 1 carries no SAMD_RUN flag: the new state's word is fetched unconditionally -- 90 % of all states start a
        // run of >= 2 (scripts/walk_chain_sim.py), and entering a run without its word costs the node load of its first state on top
        // (0.361 ms per launch against 0.374)
        if (nx >= 0) { idx = nx; len += 1; if (nx > 0) cw = chain_load(S, idx); return visited; }
        idx = w0.x; hopped = true;
        if (idx == 0) len = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the same transition over EDGE BLOCKS and HOT WORDS (samd_common.h).  What differs from st_transfer_chain_impl is only WHERE the
// facts come from -- the states visited, their order, the lengths and the count are transfer_state's (static_sam.py:98-107):
//   * the cursor's own state: its block (reference known from the entry that led here: one probe), or its hot word (then its block);
//   * every hop of the climb: ONE request -- the fail header that came with the last probe / hot word says what the link is (root,
//     root child, hub, plain state), where to look and what the match length becomes there;
//   * a miss at a hub needs nothing more: the header rode in the slot that ended the probe.
// Requires the chain words and the bigram table (the derivation builds blocks only with both).
// ------------------------------------------------------------------------------------------------
struct FailHdr { unsigned kind, ref, len; bool len_ok; };
__device__ __forceinline__ FailHdr fail_of_slot(const StaticDev &S, const uint4 &e) {
    FailHdr f;
    f.kind = (e.y >> SAMD_EB_KIND_SHIFT) & 3u; f.ref = e.w;
    const unsigned lmax = (1u << (32 - S.eb_tok_bits)) - 1u;
    f.len = e.x >> S.eb_tok_bits; f.len_ok = f.len != lmax;
    return f;
}
__device__ __forceinline__ FailHdr fail_of_hot(const uint4 &h) {
    FailHdr f;
    f.kind = (h.y >> 27) & 3u; f.ref = h.x; f.len = h.y & SAMD_EB_IDX_MASK; f.len_ok = true;
    return f;
}
// probe the block `ref` for tok: e = the slot already loaded at position p (relative to the block).  On return e is the slot that ended the
// probe -- the hit, an empty slot, or a home slot that says none of its keys lives elsewhere -- and carries the block's fail header either way.
__device__ __forceinline__ bool block_find(const StaticDev &S, unsigned ref, int tok, uint32_t p, uint4 &e) {
    const unsigned tmask = (1u << S.eb_tok_bits) - 1u, base = samd_eb_base(ref), bmask = samd_eb_mask(ref);
    const bool in_range = (unsigned)tok < tmask;
    for (uint32_t probes = 0; probes <= bmask; probes++) {
        const unsigned t = e.x & tmask;
        if (in_range && t == (unsigned)tok) return true;
        if (t == tmask) return false;
        if (probes == 0 && !(e.y & SAMD_EB_DISPLACED)) return false;      // no key of this home slot lives elsewhere: conclusive (samd_common.h)
        p = (p + 1) & bmask; e = S.blocks[base + p];
    }
    return false;
}
// Built, measured and not kept (profiles/r06_walk.md): (a) requesting a state's hot word one token early (the compiler waits for every
// outstanding load at the top of a token, so the early request became a round of its own: 0.412 vs 0.367 ms on the Zipf corpus); (b) the whole
// non-register path as ONE loop with one load instruction per iteration and a mode per lane (0.378 vs 0.349 ms: the launch is bound by the
// instructions a wave issues for ANY lane's path as much as by its dependent rounds, and the mode machine adds to every iteration).
template <int W>
__device__ __forceinline__ int st_transfer_blocks_impl(const StaticDev &S, const uint32_t *bits, int &idx, int &len, int tok, int ptok, ChainWord &cw) {
    constexpr unsigned LOW = W == 8 ? 0x7FFFu : 0x7FFFFFFFu, HI = LOW + 1u;
    if (tok < 0) { idx = 0; len = 0; cw = chain_none(); return 1; }
    const unsigned ent = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
    const bool is_tok = (ent & LOW) != LOW;
    if (is_tok && (ent & LOW) == (unsigned)tok) {            // register path: identical to st_transfer_chain_impl
        idx += 1; len += 1; cw.hub = 0;
        if (W == 8) { cw.lo = (cw.lo >> 16) | (cw.hi << 48); cw.hi = (cw.hi >> 16) | (0xFFFFull << 48); }
        else { cw.lo = (cw.lo >> 32) | (cw.hi << 32); cw.hi = (cw.hi >> 32) | (0xFFFFFFFFull << 32); }
        if (++cw.used == W) {
            if (cw.have_next) { cw.lo = cw.nlo; cw.hi = cw.nhi; cw.nlo = cw.nhi = ~0ull; cw.used = 0; cw.have_next = 0; }
            else { cw = chain_load(S, idx); return 1; }
        }
        const unsigned e2 = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
        if (cw.used == W - 1 && (e2 & LOW) != LOW) {
            const uint4 c = S.chain[idx + 1];
            cw.nlo = (unsigned long long)c.x | ((unsigned long long)c.y << 32);
            cw.nhi = (unsigned long long)c.z | ((unsigned long long)c.w << 32);
            cw.have_next = 1;
        }
        return 1;
    }
    // flagged chain entry: the cursor's state has one edge (not `tok`) and its link is the root child of ptok
    const bool climbing = is_tok && !(ent & HI) && ptok >= 0;
    const unsigned my_ref = idx > 0 ? cw.hub : 0u;           // this state's block, when the entry that led here named it
    cw = chain_none();
    int visited = climbing ? 1 : 0;
    const bool probing = climbing || st_on_child(idx);
    if (!probing && idx == 0) { st_from_root(S, bits, tok, idx, len); return 1; }

    // landing on `d` through a state's most frequent edge: its chain word when it starts a run
    auto land_e0 = [&](int d, bool run) {
        idx = d; len += 1;
        if (run) cw = chain_load(S, idx);
    };
    // what a hit in a block installs
    auto follow_block = [&](const uint4 &e) {
        len += 1;
        if (e.y & SAMD_EB_ROOTCHILD) { idx = st_child_of(tok); return; }           // its edges live in the bigram table
        idx = (int)(e.y & SAMD_EB_IDX_MASK);
        if (e.y & SAMD_EB_HUB) cw.hub = e.z; else cw = chain_from_edge<W>(e.z);
    };
    // one bigram probe under (a, tok), conclusive: the child of a visited (the caller counted it), then either the edge or the root
    auto bigram = [&](int a, uint4 e, uint32_t h, bool set_len) -> int {
        bool hit = false;
        for (uint32_t probes = 0; probes <= S.bigram_mask; probes++) {
            if (W == 8) hit = (e.x & 0x3FFFFFFFu) == ((unsigned)a | ((unsigned)tok << 15)) && tok < 0x8000;
            else hit = (e.x & 0x7FFFFFFFu) == (unsigned)a && (e.y & 0x7FFFFFFFu) == (unsigned)tok;
            if (hit || e.x == 0xFFFFFFFFu) break;
            if (probes == 0 && !((W == 8 ? e.y : e.z) & SAMD_BG_DISPLACED)) break;       // conclusive miss at the home slot (samd_common.h)
            h = (h + 1) & S.bigram_mask; e = S.bigram[h];
        }
        if (hit) {
            if (set_len) {
                const unsigned lb = W == 8 ? e.x >> 30 : (e.x >> 31) | ((e.y >> 31) << 1);
                len = lb < 3 ? (int)lb + 1 : (int)S.root16[a].w;
            }
            len += 1;
            const unsigned d = W == 8 ? e.y : e.z, extra = W == 8 ? e.z : e.w;
            if (d & SAMD_EB_ROOTCHILD) { idx = st_child_of(tok); return 0; }
            idx = (int)(d & SAMD_EB_IDX_MASK);
            if (d & 0x80000000u) cw.hub = extra;
            else if (W == 8) cw = chain_half<W>(e.z, e.w);
            else cw = chain_first<W>(e.w);
            return 0;
        }
        st_from_root(S, bits, tok, idx, len);                 // no edge: the child's suffix link is the root -- visited too
        return 1;
    };
    // THE FIRST LOAD OF EVERY LANE IS ONE INSTRUCTION (a lock-step wave pays dependent phases, not loads): a bigram slot, a slot of the
    // cursor's block, or the cursor's hot word
    const int a0 = climbing ? ptok : -2 - idx;
    const uint32_t hb = probing ? samd_bigram_hash(a0, tok) & S.bigram_mask : 0u;
    const uint32_t p0 = my_ref ? samd_eb_hash(tok) & samd_eb_mask(my_ref) : 0u;
    const uint4 *addr = probing ? S.bigram + hb : (my_ref ? S.blocks + samd_eb_base(my_ref) + p0 : S.hot + (idx > 0 ? idx : 0));
    const uint4 first = *addr;
    if (probing) {
        visited += 1;                                          // the root child itself
        return visited + bigram(a0, first, hb, climbing);
    }
    visited++;                                                 // the cursor's own state
    FailHdr f;
    if (my_ref) {                                              // a known hub: the probe decides
        uint4 e = first;
        if (block_find(S, my_ref, tok, p0, e)) { follow_block(e); return visited; }
        f = fail_of_slot(S, e);
    } else {
        const uint4 h = first;
        f = fail_of_hot(h);
        // its most frequent edge (a single state's only one) is in the word: one round for it, as through node word 0
        if ((int)h.z == tok) { land_e0((int)h.w, (h.y & SAMD_RUN) != 0); return visited; }
        if (!(h.y & SAMD_SINGLE)) {
            if (f.kind == SAMD_FK_ROOT) {                      // a branching ROOT CHILD met by index: its edges are in the bigram table
                const int a = (int)h.x;
                const uint32_t h2 = samd_bigram_hash(a, tok) & S.bigram_mask;
                return visited + bigram(a, S.bigram[h2], h2, false);
            }
            const unsigned ref = h.x;                          // a hub met by index: its word names its block, whose slots carry its fail header
            const uint32_t p = samd_eb_hash(tok) & samd_eb_mask(ref);
            uint4 e = S.blocks[samd_eb_base(ref) + p];
            if (block_find(S, ref, tok, p, e)) { follow_block(e); return visited; }
            f = fail_of_slot(S, e);
        }
    }
    // ---- the climb (static_sam.py:99-101): f describes the suffix link of the state just examined; one request per hop --------------
    for (;;) {
        visited++;
        if (f.kind == SAMD_FK_ROOT) { len = 0; st_from_root(S, bits, tok, idx, len); return visited; }
        if (f.kind == SAMD_FK_ROOTCHILD) {
            const int a = (int)f.ref;
            const uint32_t h2 = samd_bigram_hash(a, tok) & S.bigram_mask;
            return visited + bigram(a, S.bigram[h2], h2, true);
        }
        len = (int)f.len;                                      // length <- states[link].length (a HUB header's always fits: the derivation
        if (f.kind == SAMD_FK_HUB) {                           //  demotes a hub whose length overflows the slot field to STATE kind)
            const unsigned ref = f.ref;
            const uint32_t p = samd_eb_hash(tok) & samd_eb_mask(ref);
            uint4 e = S.blocks[samd_eb_base(ref) + p];
            if (block_find(S, ref, tok, p, e)) { follow_block(e); return visited; }
            f = fail_of_slot(S, e);
            continue;
        }
        // SAMD_FK_STATE: a plain state index -- its hot word is its only edge + its own fail header (or, for a hub the derivation demoted
        // to this kind, its block reference)
        const int p_idx = (int)f.ref;
        const uint4 h = S.hot[p_idx];
        if (!f.len_ok) len = S.nodes[p_idx].length & SAMD_LEN_MASK;
        if ((int)h.z == tok) { land_e0((int)h.w, (h.y & SAMD_RUN) != 0); return visited; }
        if (h.y & SAMD_SINGLE) f = fail_of_hot(h);
        else {                                                 // (a hub the derivation demoted to STATE kind: its word names its block)
            const unsigned ref = h.x;
            const uint32_t p = samd_eb_hash(tok) & samd_eb_mask(ref);
            uint4 e = S.blocks[samd_eb_base(ref) + p];
            if (block_find(S, ref, tok, p, e)) { follow_block(e); return visited; }
            f = fail_of_slot(S, e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6, last: the block path with DECOUPLED LANES (k_static_walk_async).  A lock-step wave pays, at every token, the climb of its slowest
// lane -- own state, one or two hubs, the bigram probe: three to four dependent rounds -- and with a few per cent of mismatching tokens
// nearly every wave-token has such a lane (scripts/walk_noise_probe.py: Zipf 0.135 ms per launch without mismatches, 0.220 at 2 %, 0.305 at
// 10 %).  Here a lane keeps its OWN token index: every trip of the loop is one dependent round for the whole wave -- a lane whose load is
// pending works on its result (hit: the token is done; miss: the next slot / the next hop is requested), a lane that is free starts its
// next token (register path: done in the same trip) -- so a wave needs max over lanes of (sum of a lane's rounds) trips instead of
// sum over tokens of (max over lanes of rounds).  The per-lane state machine below is st_transfer_blocks_impl taken apart at its loads;
// states visited, their order, lengths and the count are unchanged (same tests, same traces).
// ------------------------------------------------------------------------------------------------
enum { WM_READY = 0, WM_BIGRAM = 1, WM_BLOCK = 2, WM_HOT = 3 };
struct WalkLane {
    int idx, len, tok, mode, a, vtok;
    unsigned ref, pos, probes;
    bool set_len;
    ChainWord cw;
};
// the address of a lane's pending load
__device__ __forceinline__ const uint4 *wl_addr(const StaticDev &S, const WalkLane &L) {
    return L.mode == WM_BIGRAM ? S.bigram + L.pos : (L.mode == WM_BLOCK ? S.blocks + samd_eb_base(L.ref) + L.pos : S.hot + L.ref);
}
// start the token L.tok at the cursor: true = the transition is complete (register path, root, negative token); false = L.mode / ref / pos
// name the first load
template <int W>
__device__ __forceinline__ bool wl_start(const StaticDev &S, const uint32_t *bits, WalkLane &L) {
    constexpr unsigned LOW = W == 8 ? 0x7FFFu : 0x7FFFFFFFu, HI = LOW + 1u;
    ChainWord &cw = L.cw;
    const int tok = L.tok;
    L.vtok = 1;
    if (tok < 0) { L.idx = 0; L.len = 0; cw = chain_none(); return true; }
    const unsigned ent = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
    const bool is_tok = (ent & LOW) != LOW;
    if (is_tok && (ent & LOW) == (unsigned)tok) {            // register path: identical to st_transfer_blocks_impl
        L.idx += 1; L.len += 1; cw.hub = 0;
        if (W == 8) { cw.lo = (cw.lo >> 16) | (cw.hi << 48); cw.hi = (cw.hi >> 16) | (0xFFFFull << 48); }
        else { cw.lo = (cw.lo >> 32) | (cw.hi << 32); cw.hi = (cw.hi >> 32) | (0xFFFFFFFFull << 32); }
        if (++cw.used == W) {
            if (cw.have_next) { cw.lo = cw.nlo; cw.hi = cw.nhi; cw.nlo = cw.nhi = ~0ull; cw.used = 0; cw.have_next = 0; }
            else { const int ptok = cw.ptok; cw = chain_load(S, L.idx); cw.ptok = ptok; return true; }
        }
        const unsigned e2 = W == 8 ? (unsigned)(cw.lo & 0xFFFFull) : (unsigned)(cw.lo & 0xFFFFFFFFull);
        if (cw.used == W - 1 && (e2 & LOW) != LOW) {
            const uint4 c = S.chain[L.idx + 1];
            cw.nlo = (unsigned long long)c.x | ((unsigned long long)c.y << 32);
            cw.nhi = (unsigned long long)c.z | ((unsigned long long)c.w << 32);
            cw.have_next = 1;
        }
        return true;
    }
    const int ptok = cw.ptok;
    const bool climbing = is_tok && !(ent & HI) && ptok >= 0;
    const unsigned my_ref = L.idx > 0 ? cw.hub : 0u;
    cw = chain_none();
    if (!climbing && L.idx == 0) { st_from_root(S, bits, tok, L.idx, L.len); return true; }
    L.probes = 0u;
    if (climbing || st_on_child(L.idx)) {
        L.vtok = (climbing ? 1 : 0) + 1;                       // (the flagged state,) the root child
        L.mode = WM_BIGRAM; L.a = climbing ? ptok : -2 - L.idx; L.set_len = climbing;
        L.pos = samd_bigram_hash(L.a, tok) & S.bigram_mask;
    } else if (my_ref) {
        L.mode = WM_BLOCK; L.ref = my_ref; L.pos = samd_eb_hash(tok) & samd_eb_mask(my_ref);
    } else {
        L.mode = WM_HOT; L.ref = (unsigned)L.idx;
    }
    return false;
}
// the pending load of L has landed in e: true = the transition is complete; false = the next load is named
template <int W>
__device__ __forceinline__ bool wl_step(const StaticDev &S, const uint32_t *bits, WalkLane &L, const uint4 &e) {
    constexpr unsigned LOW = W == 8 ? 0x7FFFu : 0x7FFFFFFFu;
    ChainWord &cw = L.cw;
    const int tok = L.tok;
    const unsigned tmask = (1u << S.eb_tok_bits) - 1u;
    FailHdr f;
    if (L.mode == WM_BIGRAM) {
        bool hit;
        if (W == 8) hit = (e.x & 0x3FFFFFFFu) == ((unsigned)L.a | ((unsigned)tok << 15)) && tok < 0x8000;
        else hit = (e.x & 0x7FFFFFFFu) == (unsigned)L.a && (e.y & 0x7FFFFFFFu) == (unsigned)tok;
        const unsigned d = W == 8 ? e.y : e.z, extra = W == 8 ? e.z : e.w;
        if (hit) {
            if (L.set_len) {
                const unsigned lb = W == 8 ? e.x >> 30 : (e.x >> 31) | ((e.y >> 31) << 1);
                L.len = lb < 3 ? (int)lb + 1 : (int)S.root16[L.a].w;
            }
            L.len += 1;
            if (d & SAMD_EB_ROOTCHILD) { L.idx = st_child_of(tok); return true; }
            L.idx = (int)(d & SAMD_EB_IDX_MASK);
            if (d & 0x80000000u) cw.hub = extra;
            else if (W == 8) cw = chain_half<W>(e.z, e.w);
            else cw = chain_first<W>(e.w);
            return true;
        }
        if (e.x == 0xFFFFFFFFu || (L.probes == 0 && !(d & SAMD_BG_DISPLACED)) || L.probes >= S.bigram_mask) {
            st_from_root(S, bits, tok, L.idx, L.len);          // no such edge: the child's link is the root -- visited too
            L.vtok += 1;
            return true;
        }
        L.pos = (L.pos + 1) & S.bigram_mask; L.probes++;
        return false;
    }
    if (L.mode == WM_BLOCK) {
        const unsigned t = e.x & tmask;
        if ((unsigned)tok < tmask && t == (unsigned)tok) {     // the edge
            L.len += 1;
            if (e.y & SAMD_EB_ROOTCHILD) { L.idx = st_child_of(tok); return true; }
            L.idx = (int)(e.y & SAMD_EB_IDX_MASK);
            if (e.y & SAMD_EB_HUB) cw.hub = e.z; else cw = chain_from_edge<W>(e.z);
            return true;
        }
        if (!(t == tmask || (L.probes == 0 && !(e.y & SAMD_EB_DISPLACED)) || L.probes >= samd_eb_mask(L.ref))) {
            L.pos = (L.pos + 1) & samd_eb_mask(L.ref); L.probes++;
            return false;
        }
        f = fail_of_slot(S, e);
    } else {                                                   // WM_HOT: the word of state L.ref
        if ((int)e.z == tok) {                                 // its most frequent edge (a single state's only one)
            L.idx = (int)e.w; L.len += 1;
            if (e.y & SAMD_RUN) cw = chain_load(S, L.idx);
            return true;
        }
        f = fail_of_hot(e);
        if (!(e.y & SAMD_SINGLE)) {
            L.probes = 0u;
            if (f.kind == SAMD_FK_ROOT) {                      // a branching ROOT CHILD met by index: its edges are in the bigram table
                L.mode = WM_BIGRAM; L.a = (int)e.x; L.set_len = false; L.pos = samd_bigram_hash(L.a, tok) & S.bigram_mask;
            } else {                                           // a hub: its word names its block (same state: nothing new visited)
                L.mode = WM_BLOCK; L.ref = e.x; L.pos = samd_eb_hash(tok) & samd_eb_mask(L.ref);
            }
            return false;
        }
    }
    // ---- the state just examined has no edge on tok: hop to its suffix link (static_sam.py:99-101), which f describes --------------------
    L.vtok += 1;
    if (f.kind == SAMD_FK_ROOT) { L.len = 0; st_from_root(S, bits, tok, L.idx, L.len); return true; }
    L.probes = 0u;
    if (f.kind == SAMD_FK_ROOTCHILD) {
        L.mode = WM_BIGRAM; L.a = (int)f.ref; L.set_len = true; L.pos = samd_bigram_hash(L.a, tok) & S.bigram_mask;
    } else if (f.kind == SAMD_FK_HUB) {
        L.len = (int)f.len; L.mode = WM_BLOCK; L.ref = f.ref; L.pos = samd_eb_hash(tok) & samd_eb_mask(L.ref);
    } else {
        L.len = f.len_ok ? (int)f.len : (int)(S.nodes[f.ref].length & SAMD_LEN_MASK);
        L.mode = WM_HOT; L.ref = f.ref;
    }
    return false;
}

// n committed transitions of one cursor (transfer_tokens, static_sam.py:118-120) -- the single-wavefront kernels' form: all lanes
// run the same cursor (uniform addresses: one request per load), with the chain words and the root-child hash when the automaton
// has them, so that a run of tokens that follows the corpus costs one load per 8 (4) tokens instead of one per token
__device__ __forceinline__ void st_transfer_tokens(const StaticDev &S, int &idx, int &len, const int *toks, int n) {
    if (S.chain && S.chain_w == 8) {
        ChainWord cw = chain_none();
        for (int i = 0; i < n; i++) st_transfer_chain<8>(S, S.rc_bits, idx, len, toks[i], cw);
        idx = st_resolve(S, idx);
    } else if (S.chain) {
        ChainWord cw = chain_none();
        for (int i = 0; i < n; i++) st_transfer_chain<4>(S, S.rc_bits, idx, len, toks[i], cw);
        idx = st_resolve(S, idx);
    } else {
        for (int i = 0; i < n; i++) st_transfer(S, idx, len, toks[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// dynamic automaton (single wavefront, uniform control flow)
// reference: samd_sam_only/sam/dyn_sam.py:50-114 (identical in samd/sam/dyn_sam.py:41-97)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t dyn_key(int state, int tok) { return ((uint64_t)(uint32_t)state << 32) | (uint32_t)tok; }
__device__ __forceinline__ uint32_t dyn_hash(uint64_t k) {
    k *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(k >> 29);
}

// `tok in states[state].next` : 64 consecutive slots per probe round, one ballot decides.
// returns the slot (== edge id) or -1; *free_slot = first empty slot of the probe sequence.
__device__ __forceinline__ int dyn_find(const SessionDev &D, int state, int tok, int *free_slot) {
    const uint64_t key = dyn_key(state, tok);
    uint32_t h = dyn_hash(key) & D.hmask;
    const int lane = lane_id();
    for (uint32_t round = 0; round <= D.hmask; round += WAVE) {
        const uint32_t slot = (h + lane) & D.hmask;
        const uint64_t k = D.hkey[slot];
        const unsigned long long hit = __ballot(k == key);
        const unsigned long long empty = __ballot(k == SAMD_HEMPTY);
        if (hit) return (int)((h + (uint32_t)(__ffsll((long long)hit) - 1)) & D.hmask);
        if (empty) { *free_slot = (int)((h + (uint32_t)(__ffsll((long long)empty) - 1)) & D.hmask); return -1; }
        h = (h + WAVE) & D.hmask;
    }
    *free_slot = -1;
    return -1;
}

// states[state].next[tok] = dst for a new key, appended to the state's dict order.  tail = D.tail[state] as the caller read it
// (together with the probe that found `slot` free: one memory round trip instead of two)
__device__ __forceinline__ void dyn_insert(const SessionDev &D, int slot, int state, int tok, int dst, int tail) {
    if (lane_id() == 0) {
        D.hkey[slot] = dyn_key(state, tok); D.hdst[slot] = dst; D.hnext[slot] = -1;
        if (tail < 0) D.head[state] = slot; else D.hnext[tail] = slot;
        D.tail[state] = slot;
    }
    wave_mem_sync();
}
__device__ __forceinline__ void dyn_insert(const SessionDev &D, int slot, int state, int tok, int dst) {
    dyn_insert(D, slot, state, tok, dst, D.tail[state]);
}

// transfer_state (dyn_sam.py:78-87).  Everything whose address follows from the current state goes out with the probe of its edges:
// its suffix link (where a failed probe continues), and -- one hop later, with the NEXT probe -- that state's length.  One memory round
// trip per visited state where the plain form paid three (probe, link, length); the session's arena is L2-resident, ~0.6 us each.
__device__ __forceinline__ void dyn_transfer(const SessionDev &D, int &idx, int &len, int tok) {
    int fs, e;
    bool hopped = false;
    for (;;) {
        const int lk = idx != 0 ? D.link[idx] : -1;              // requested with the probe below
        const int ln = hopped ? D.length[idx] : 0;
        e = dyn_find(D, idx, tok, &fs);
        if (hopped) len = ln;
        if (e >= 0 || idx == 0) break;
        idx = lk; hopped = true;
    }
    if (e >= 0) { idx = D.hdst[e]; len += 1; } else { idx = 0; len = 0; }
}

struct DynRegs { int n_states, n_edges, last, max_length, error; };

__device__ __forceinline__ int dyn_new_state(const SessionDev &D, DynRegs &R, int link, int length, int minend) {
    const int s = R.n_states++;
    if (lane_id() == 0) { D.link[s] = link; D.length[s] = length; D.minend[s] = minend; D.head[s] = -1; D.tail[s] = -1; }
    return s;
}

// add_state (dyn_sam.py:50-76)
__device__ __forceinline__ void dyn_add_state(const SessionDev &D, DynRegs &R, int tok) {
    if (R.n_states + 2 > D.cap_states || (uint32_t)(R.n_edges + 8) * 2 > D.hmask) { R.error = SAMD_E_CAPACITY; return; }
    R.max_length += 1;
    const int cur = dyn_new_state(D, R, -1, R.max_length, R.max_length);
    wave_mem_sync();
    int p = R.last, e = -1, fs = -1, len_p = 0;
    while (p != -1) {
        // the suffix link, the dict tail and the length of p are requested with the probe of its edges: one round trip per chain hop,
        // not three
        const int lk = D.link[p], tl = D.tail[p];
        len_p = D.length[p];
        e = dyn_find(D, p, tok, &fs);
        if (e >= 0) break;
        if (fs < 0 || (uint32_t)(R.n_edges + 8) * 2 > D.hmask) { R.error = SAMD_E_CAPACITY; return; }
        dyn_insert(D, fs, p, tok, cur, tl); R.n_edges++;
        p = lk;
    }
    if (p == -1) {
        if (lane_id() == 0) D.link[cur] = 0;
    } else {
        const int q = D.hdst[e];
        if (len_p + 1 == D.length[q]) {
            if (lane_id() == 0) D.link[cur] = q;
        } else {
            // clone = deepcopy(q): same dict (order kept), link, min_endpos; length = len(p)+1
            const int clone = dyn_new_state(D, R, D.link[q], len_p + 1, D.minend[q]);
            wave_mem_sync();
            for (int qe = D.head[q]; qe >= 0; qe = D.hnext[qe]) {
                const int t = (int)(uint32_t)D.hkey[qe];
                const int d = D.hdst[qe];
                (void)dyn_find(D, clone, t, &fs);
                if ((uint32_t)(R.n_edges + 8) * 2 > D.hmask || fs < 0) { R.error = SAMD_E_CAPACITY; return; }
                dyn_insert(D, fs, clone, t, d); R.n_edges++;
            }
            while (p != -1) {
                const int pe = dyn_find(D, p, tok, &fs);
                if (pe < 0 || D.hdst[pe] != q) break;
                if (lane_id() == 0) D.hdst[pe] = clone;      // re-point: dict order unchanged
                p = D.link[p];
            }
            if (lane_id() == 0) { D.link[q] = clone; D.link[cur] = clone; }
        }
    }
    R.last = cur;
    wave_mem_sync();
}

__device__ __forceinline__ void dyn_load(const SessionDev &D, DynRegs &R) {
    R.n_states = D.meta[M_NSTATES]; R.n_edges = D.meta[M_NEDGES]; R.last = D.meta[M_LAST];
    R.max_length = D.meta[M_MAXLEN]; R.error = D.meta[M_ERROR];
}
__device__ __forceinline__ void dyn_store(const SessionDev &D, const DynRegs &R) {
    if (lane_id() == 0) {
        D.meta[M_NSTATES] = R.n_states; D.meta[M_NEDGES] = R.n_edges; D.meta[M_LAST] = R.last;
        D.meta[M_MAXLEN] = R.max_length; D.meta[M_ERROR] = R.error;
    }
}

// DynSAM.add_tokens (dyn_sam.py:101-105): per token transfer the cursor FIRST, then extend; the tokens
// are appended to input_ids afterwards.  `toks` may live in global memory or LDS.
__device__ __forceinline__ void dyn_add_tokens(const SessionDev &D, const int *toks, int n) {
    DynRegs R; dyn_load(D, R);
    int ci = D.meta[M_CUR_IDX], cl = D.meta[M_CUR_LEN], nt = D.meta[M_NTEXT];
    if (nt + n > D.cap_text) { R.error = SAMD_E_CAPACITY; n = 0; }
    for (int i = 0; i < n && R.error == 0; i++) {
        const int t = toks[i];
        dyn_transfer(D, ci, cl, t);
        dyn_add_state(D, R, t);
    }
    if (R.error == 0) {
        for (int i = lane_id(); i < n; i += WAVE) D.text[nt + i] = toks[i];
        nt += n;
    }
    dyn_store(D, R);
    if (lane_id() == 0) { D.meta[M_CUR_IDX] = ci; D.meta[M_CUR_LEN] = cl; D.meta[M_NTEXT] = nt; }
    wave_mem_sync();
}

// ------------------------------------------------------------------------------------------------
// drafts
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int draft_size(int match, double alpha, int max_predicts) {
    // n = min(max_predicts, 1 + int(match_length * alpha))   (dyn_sam.py:117, static_sam.py:183)
    int n = 1 + (int)((double)match * alpha);
    n = n < max_predicts ? n : max_predicts;
    return n > SAMD_MAX_DRAFT ? SAMD_MAX_DRAFT : n;
}

// LDS scratch of the single-wavefront kernels
struct StepShared {
    // draft under construction
    int tokens[SAMD_MAX_DRAFT];
    int parent[SAMD_MAX_DRAFT];
    int position[SAMD_MAX_DRAFT];
    unsigned long long mask[SAMD_MAX_DRAFT], mask_hi[SAMD_MAX_DRAFT];   // ancestors among nodes 0..63 / 64..127
    __attribute__((aligned(16))) unsigned char path[SAMD_MAX_DRAFT][SAMD_MAX_DRAFT];   // retrieve rows, PATH_PAD padded
    int node_argmax[SAMD_MAX_DRAFT];
    int accepted[SAMD_MAX_DRAFT];
    unsigned long long child_mask[2];
    int red[4];
    // best-first search heap (static_sam.py:184-214): <= 1 + 8*(n-1) live items
    double h_prob[8 * SAMD_MAX_DRAFT + 8];
    int h_tok[8 * SAMD_MAX_DRAFT + 8], h_idx[8 * SAMD_MAX_DRAFT + 8], h_anc[8 * SAMD_MAX_DRAFT + 8], h_dep[8 * SAMD_MAX_DRAFT + 8];
    int dep_cnt[SAMD_MAX_DRAFT + 2];
    int c_tok[SAMD_TOPK], c_dst[SAMD_TOPK], c_cnt[SAMD_TOPK];
    int pop_ok, pop_tok, pop_idx, pop_anc, pop_dep; double pop_prob;
};

// PATH_PAD into the first `rows` retrieve rows (16-byte stores: the 16 KB table is never filled whole -- a draft has few leaf rows)
__device__ __forceinline__ void path_fill(StepShared &sh, int rows) {
    const uint4 pad = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    static_assert(PATH_PAD == 255 && SAMD_MAX_DRAFT % 16 == 0, "");
    uint4 *p = reinterpret_cast<uint4 *>(&sh.path[0][0]);
    for (int k = lane_id(); k < rows * (SAMD_MAX_DRAFT / 16); k += WAVE) p[k] = pad;
}

// sequence draft [start] + input_ids[e+1 : e+n] from `text` (dyn_sam.py:118-119); returns its length
__device__ __forceinline__ int seq_draft_var(StepShared &sh, const int *text, int n_text, int endpos, int n, int start) {
    const int lo = endpos + 1;
    int hi = endpos + n; hi = hi > n_text ? n_text : hi;
    const int m = 1 + (hi > lo ? hi - lo : 0);
    for (int i = lane_id(); i < m; i += WAVE) { sh.tokens[i] = (i == 0) ? start : text[endpos + i]; sh.parent[i] = i - 1; }   // (two nodes per lane above 64)
    __syncthreads();
    return m;
}

// fixed-length, zero padded (samd/sam/dyn_sam.py:107-113, samd/sam/static_sam.py:119-125)
__device__ __forceinline__ int seq_draft_fixed(StepShared &sh, const int *text, int n_text, int endpos, int n_predicts, int start) {
    for (int i = lane_id(); i < n_predicts; i += WAVE) {
        int v = 0;
        if (i == 0) v = start; else if (endpos + i < n_text) v = text[endpos + i];
        sh.tokens[i] = v; sh.parent[i] = i - 1;
    }
    __syncthreads();
    return n_predicts;
}

// CPython heapq on SearchItem(prob) (static_sam.py:13-19): lane 0 only
__device__ __forceinline__ void hq_set(StepShared &sh, int pos, double p, int t, int i, int a, int d) {
    sh.h_prob[pos] = p; sh.h_tok[pos] = t; sh.h_idx[pos] = i; sh.h_anc[pos] = a; sh.h_dep[pos] = d;
}
__device__ __forceinline__ void hq_move(StepShared &sh, int dst, int src) {
    sh.h_prob[dst] = sh.h_prob[src]; sh.h_tok[dst] = sh.h_tok[src]; sh.h_idx[dst] = sh.h_idx[src];
    sh.h_anc[dst] = sh.h_anc[src]; sh.h_dep[dst] = sh.h_dep[src];
}
__device__ __forceinline__ void hq_siftdown(StepShared &sh, int start, int pos, double p, int t, int i, int a, int d) {
    while (pos > start) {                         // _siftdown: move parents down while new < parent
        const int par = (pos - 1) >> 1;
        if (p < sh.h_prob[par]) { hq_move(sh, pos, par); pos = par; continue; }
        break;
    }
    hq_set(sh, pos, p, t, i, a, d);
}
__device__ __forceinline__ void hq_push(StepShared &sh, int &hn, double p, int t, int i, int a, int d) {
    hn++;
    hq_siftdown(sh, 0, hn - 1, p, t, i, a, d);
}
__device__ __forceinline__ void hq_pop(StepShared &sh, int &hn, double &p, int &t, int &i, int &a, int &d) {
    hn--;
    const double lp = sh.h_prob[hn]; const int lt = sh.h_tok[hn], li = sh.h_idx[hn], la = sh.h_anc[hn], ld = sh.h_dep[hn];
    if (hn == 0) { p = lp; t = lt; i = li; a = la; d = ld; return; }
    p = sh.h_prob[0]; t = sh.h_tok[0]; i = sh.h_idx[0]; a = sh.h_anc[0]; d = sh.h_dep[0];
    int pos = 0, child = 1;                       // _siftup: to a leaf, right child when `not left < right`
    while (child < hn) {
        const int right = child + 1;
        if (right < hn && !(sh.h_prob[child] < sh.h_prob[right])) child = right;
        hq_move(sh, pos, child); pos = child; child = 2 * pos + 1;
    }
    hq_siftdown(sh, 0, pos, lp, lt, li, la, ld);
}

// StaticSAM.gen_draft (static_sam.py:182-215): best-first tree of <= n nodes, <= K per depth
__device__ __forceinline__ int tree_draft(StepShared &sh, const StaticDev &S, int index, int n, int K, int start) {
    const int lane = lane_id();
    int hn = 0, m = 0;
    for (int k = lane; k < SAMD_MAX_DRAFT + 2; k += WAVE) sh.dep_cnt[k] = 0;
    __syncthreads();
    if (lane == 0) hq_push(sh, hn, -1.0, start, index, -1, 0);
    __syncthreads();
    while (m != n) {
        if (lane == 0) {
            int ok = 0; double p = 0; int t = 0, i = 0, a = 0, d = 0;
            while (hn > 0) {
                hq_pop(sh, hn, p, t, i, a, d);
                if (sh.dep_cnt[d] + 1 > K) continue;          // this depth is full: drop without expanding
                sh.dep_cnt[d] += 1; ok = 1; break;
            }
            sh.pop_ok = ok; sh.pop_prob = p; sh.pop_tok = t; sh.pop_idx = i; sh.pop_anc = a; sh.pop_dep = d;
        }
        __syncthreads();
        if (!sh.pop_ok) break;
        const int cur = m;
        if (lane == 0) { sh.tokens[m] = sh.pop_tok; sh.parent[m] = sh.pop_anc; }
        m++;
        if (m == n) break;
        // expand: the first min(K, 8, deg) successors in top-k order, one lane each
        // everything whose address is known from the popped state goes out in ONE round trip: the node's words, this lane's inline edge
        // and its child count from the top-k count table (samd_common.h); only a state of degree > 5 needs a second one (spill head)
        const int4 *np = reinterpret_cast<const int4 *>(S.nodes + sh.pop_idx);
        const int4 w1 = np[1];                                      // {aux = cnt_endpos, deg, e1}
        const int le = lane < SAMD_INLINE_EDGES ? lane : 0;
        const int2 ie = *reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(np) + SAMD_EDGE_WORD(le));
        const int sp = reinterpret_cast<const int *>(np)[14];
        const int cnt_pre = S.topk_cnt ? S.topk_cnt[(size_t)sh.pop_idx * SAMD_TOPK + (lane & (SAMD_TOPK - 1))] : 0;
        int kk = w1.y < SAMD_TOPK ? w1.y : SAMD_TOPK; kk = kk < K ? kk : K;
        if (lane < kk) {
            int tk = ie.x, ds = ie.y;
            if (lane >= SAMD_INLINE_EDGES) { const SamEdge ed = S.spill[sp + lane - SAMD_INLINE_EDGES]; tk = ed.tok; ds = ed.dst; }
            sh.c_tok[lane] = tk; sh.c_dst[lane] = ds; sh.c_cnt[lane] = S.topk_cnt ? cnt_pre : S.nodes[ds].aux;
        }
        __syncthreads();
        if (lane == 0) {
            const double cnt_sum = (double)w1.x, pp = sh.pop_prob;
            const int dep = sh.pop_dep + 1;
            for (int j = 0; j < kk; j++) {
                const double n_prob = (double)sh.c_cnt[j] / cnt_sum;      // Python int / int
                hq_push(sh, hn, pp * n_prob, sh.c_tok[j], sh.c_dst[j], cur, dep);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    return m;
}

// gen_buffers (static_sam.py:148-180): depth, ancestor mask, root->leaf rows.  One lane per node.
// reverse = Token-Recycle row order (samd/tree_model/token_recycle/utils.py:88).
__device__ __forceinline__ void build_buffers(StepShared &sh, int n, int reverse, int &n_leaves, int &max_depth) {
    const int lane = lane_id();
    if (lane == 0) { sh.child_mask[0] = 0ull; sh.child_mask[1] = 0ull; }
    __syncthreads();
    // a well-formed parent array has parent[0] == -1 and 0 <= parent[i] < i; anything else (an externally supplied draft
    // with a missing ancestor) is re-attached to the root so that the ancestor walks below always terminate
    for (int i = lane; i < n; i += WAVE) {
        const int p = sh.parent[i];
        if (i == 0) sh.parent[0] = -1;
        else if (p < 0 || p >= i) sh.parent[i] = 0;
    }
    __syncthreads();
    int md = 0;
    for (int i = lane; i < n; i += WAVE) {                    // one lane per node, two rounds above 64 nodes
        int depth = 0; unsigned long long m = 0ull, mh = 0ull;
        for (int j = i; j != -1; j = sh.parent[j]) { if (j < 64) m |= 1ull << j; else mh |= 1ull << (j - 64); depth++; }
        depth -= 1;
        sh.position[i] = depth; sh.mask[i] = m; sh.mask_hi[i] = mh;
        if (i > 0) { const int p = sh.parent[i]; atomicOr(&sh.child_mask[p >> 6], 1ull << (p & 63)); }
        md = depth + 1 > md ? depth + 1 : md;
    }
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(md, o); md = v > md ? v : md; }
    __syncthreads();
    const int n0 = n < 64 ? n : 64, n1 = n > 64 ? n - 64 : 0;
    const unsigned long long valid0 = n0 >= 64 ? ~0ull : ((1ull << n0) - 1ull), valid1 = n1 >= 64 ? ~0ull : ((1ull << n1) - 1ull);
    const unsigned long long leaf0 = valid0 & ~sh.child_mask[0], leaf1 = valid1 & ~sh.child_mask[1];
    const int nl0 = __popcll(leaf0), nl = nl0 + __popcll(leaf1);
    path_fill(sh, nl);
    __syncthreads();
    for (int i = lane; i < n; i += WAVE) {
        const bool is_leaf = i < 64 ? ((leaf0 >> i) & 1ull) : ((leaf1 >> (i - 64)) & 1ull);
        if (!is_leaf) continue;
        int r = i < 64 ? __popcll(leaf0 & ((1ull << i) - 1ull)) : nl0 + __popcll(leaf1 & ((1ull << (i - 64)) - 1ull));      // leaves in increasing node index
        if (reverse) r = nl - 1 - r;
        for (int j = i; j != -1; j = sh.parent[j]) sh.path[r][sh.position[j]] = (unsigned char)j;
    }
    __syncthreads();
    n_leaves = nl; max_depth = md;
}

__device__ __forceinline__ void store_draft(const SessionDev &D, StepShared &sh, int type, int n, int n_leaves, int max_depth,
                                            int idx_dyn, int m_dyn, int idx_st, int m_st, int reverse) {
    const int i = lane_id();
    for (int k = i; k < SAMD_MAX_DRAFT; k += WAVE) {
        if (k < n) { D.tokens[k] = sh.tokens[k]; D.parent[k] = sh.parent[k]; D.position[k] = sh.position[k]; D.mask[k] = sh.mask[k]; D.mask_hi[k] = sh.mask_hi[k]; }
        else { D.tokens[k] = 0; D.parent[k] = k - 1; D.position[k] = 0; D.mask[k] = 0ull; D.mask_hi[k] = 0ull; }   // padded rows attend nothing
    }
    for (int k = i; k < n_leaves * max_depth; k += WAVE) {
        const unsigned char v = sh.path[k / max_depth][k % max_depth];
        D.retrieve[k] = v == PATH_PAD ? -1 : (int)v;
    }
    if (i == 0) {
        D.dmeta[D_TYPE] = type; D.dmeta[D_N] = n; D.dmeta[D_NLEAVES] = n_leaves; D.dmeta[D_MAXDEPTH] = max_depth;
        D.dmeta[D_IDX_DYN] = idx_dyn; D.dmeta[D_MATCH_DYN] = m_dyn; D.dmeta[D_IDX_ST] = idx_st; D.dmeta[D_MATCH_ST] = m_st;
        D.dmeta[8] = reverse;
    }
}

// DraftModel.lookup (samd_sam_only/draft.py:50-59 | samd/draft.py:52-63) + gen_buffers
__device__ __forceinline__ void do_draft(const SessionDev &D, const StaticDev &S, bool have_static, const samd_params_t &P,
                                         StepShared &sh, int start) {
    int id = D.meta[M_CUR_IDX], md = D.meta[M_CUR_LEN];
    dyn_transfer(D, id, md, start);
    int is = 0, ms = 0;
    if (have_static) { is = D.meta[M_ST_IDX]; ms = D.meta[M_ST_LEN]; st_transfer(S, is, ms, start); }
    ms -= P.len_bias;
    int type, n;
    if (P.variant == 0) {
        if (md >= ms || !have_static) {
            type = 0;
            n = seq_draft_var(sh, D.text, D.meta[M_NTEXT], D.minend[id], draft_size(md, P.alpha, P.max_predicts), start);
        } else {
            type = 1;
            n = tree_draft(sh, S, is, draft_size(ms, P.alpha, P.max_predicts), P.K, start);
        }
    } else {
        const int best = md > ms ? md : ms;
        if (best >= P.len_threshold) {
            type = 0;
            if (md >= ms || !have_static) {
                // to_anc (samd/sam/dyn_sam.py:99-105)
                int a = id;
                if (a != 0) {
                    const int maxlen = D.meta[M_MAXLEN];
                    int to_end = maxlen - D.minend[a];
                    while (D.link[a] != 0 && P.n_predicts > to_end) { a = D.link[a]; to_end = maxlen - D.minend[a]; }
                }
                n = seq_draft_fixed(sh, D.text, D.meta[M_NTEXT], D.minend[a], P.n_predicts, start);
            } else {
                n = seq_draft_fixed(sh, S.text, S.n_text, S.nodes[is].aux, P.n_predicts, start);
            }
        } else {
            type = 2; n = 1;                      // the tree model drafts (samd/draft.py:63): only the start token here
            if (lane_id() == 0) { sh.tokens[0] = start; sh.parent[0] = -1; }
            __syncthreads();
        }
    }
    int nl, mxd;
    build_buffers(sh, n, 0, nl, mxd);
    store_draft(D, sh, type, n, nl, mxd, id, md, is, ms, 0);
}

// eval_posterior, greedy (samd_sam_only/utils.py:127-141) + update_state's selection
// (samd_sam_only/samd_model.py:165-169), on the draft held in LDS (tokens, path rows).
__device__ __forceinline__ void do_accept(const SessionDev &D, StepShared &sh, const int *node_argmax, int type, int n,
                                          int n_leaves, int max_depth, int &accept_out, int &next_token_out) {
    const int i = lane_id();
    for (int k = i; k < n; k += WAVE) sh.node_argmax[k] = node_argmax[k];
    __syncthreads();
    // one lane per candidate row (two rounds above 64 leaves); first maximum over rows: key = acc * 128 + (127 - row)
    int key = -1;
    for (int row = i; row < n_leaves; row += WAVE) {
        int acc = 0;
        for (int j = 1; j < max_depth; j++) {
            const int cj = sh.path[row][j], pj = sh.path[row][j - 1];
            const int cand = cj == PATH_PAD ? 0 : sh.tokens[cj];              // pad token 0 (utils.py:95-96)
            const int am = sh.node_argmax[pj == PATH_PAD ? n - 1 : pj];       // logits[-1] = last node (samd_model.py:144)
            if (cand != am) break;
            acc++;
        }
        const int kk = acc * SAMD_MAX_DRAFT + (SAMD_MAX_DRAFT - 1 - row);
        key = kk > key ? kk : key;
    }
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(key, o); key = v > key ? v : key; }
    const int best_acc = key / SAMD_MAX_DRAFT, best = SAMD_MAX_DRAFT - 1 - (key % SAMD_MAX_DRAFT);
    const int a = best_acc + 1;
    for (int k = i; k < a; k += WAVE) {
        const int node = sh.path[best][k];
        const int tk = node == PATH_PAD ? 0 : sh.tokens[node];
        sh.accepted[k] = tk;
        D.acc_tokens[k] = tk;
        D.kv_index[k] = node == PATH_PAD ? -1 : node;
    }
    const int nn_raw = sh.path[best][best_acc];
    const int nn = nn_raw == PATH_PAD ? n - 1 : nn_raw;
    const int next_token = sh.node_argmax[nn];
    if (i == 0) {
        const int start = D.cache_length[0];
        D.verdict[V_BEST] = best; D.verdict[V_ACCEPT] = a; D.verdict[V_NEXT_NODE] = nn; D.verdict[V_NEXT_TOKEN] = next_token;
        D.verdict[V_KV_START] = start; D.verdict[V_IS_TREE] = (type != 0);
        D.cache_length[0] = start + a;                                         // cache.py:133
        D.start_token[0] = next_token;
        D.counters[C_STEPS] += 1; D.counters[C_TOKENS] += a;
        D.counters[type == 0 ? C_SEQ_STEPS : C_TREE_STEPS] += 1;
    }
    __syncthreads();
    accept_out = a; next_token_out = next_token;
}

// update_state's selection (samd_sam_only/samd_model.py:165-169) for a verdict that was decided elsewhere -- the sampling branch of
// eval_posterior (utils.py:142-184, samd_posterior_sampled*): given[0] = best candidate row, given[1] = accept length (root included),
// *next_token = the start token drawn for the next step.  Everything else is do_accept's bookkeeping.
__device__ __forceinline__ void do_accept_given(const SessionDev &D, StepShared &sh, const int *given, const int *next_token_ptr, int type, int n,
                                                int n_leaves, int max_depth, int &accept_out, int &next_token_out) {
    const int i = lane_id();
    int best = given[0], a = given[1];
    best = best < 0 ? 0 : (best >= n_leaves ? (n_leaves > 0 ? n_leaves - 1 : 0) : best);
    a = a < 1 ? 1 : (a > max_depth ? max_depth : a);
    for (int k = i; k < a; k += WAVE) {
        const int node = sh.path[best][k];
        const int tk = node == PATH_PAD ? 0 : sh.tokens[node];
        sh.accepted[k] = tk;
        D.acc_tokens[k] = tk;
        D.kv_index[k] = node == PATH_PAD ? -1 : node;
    }
    const int nn_raw = sh.path[best][a - 1];
    const int nn = nn_raw == PATH_PAD ? n - 1 : nn_raw;
    const int next_token = next_token_ptr[0];
    if (i == 0) {
        const int start = D.cache_length[0];
        D.verdict[V_BEST] = best; D.verdict[V_ACCEPT] = a; D.verdict[V_NEXT_NODE] = nn; D.verdict[V_NEXT_TOKEN] = next_token;
        D.verdict[V_KV_START] = start; D.verdict[V_IS_TREE] = (type != 0);
        D.cache_length[0] = start + a;
        D.start_token[0] = next_token;
        D.counters[C_STEPS] += 1; D.counters[C_TOKENS] += a;
        D.counters[type == 0 ? C_SEQ_STEPS : C_TREE_STEPS] += 1;
    }
    __syncthreads();
    accept_out = a; next_token_out = next_token;
}
