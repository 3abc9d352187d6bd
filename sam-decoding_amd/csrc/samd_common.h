// samd_common.h -- shared host/device layouts of libsamd_hip (gfx950 only).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/samd_hip.h"

#define SAMD_ABI_VERSION 2
#define SAMD_INLINE_EDGES 5        // edges stored inside the 64-byte node
#define SAMD_SPILL_HEAD 3          // spill block starts with top-k ranks 5,6,7 (-1 padded)

// One automaton state = one 64-byte HBM line.  Measured on MI355X (scripts/hbm_probe.hip, profiles/r01_hbm_probe.md):
// the memory system retires ~48 G requests/s whatever their size; a lane that reads its whole 64-byte node with four
// dwordx4 loads generates 1.5 requests per node (1.28 TB/s), a lane that reads ONE dwordx4 of it generates exactly one
// (47 G nodes/s).  So the first 16 bytes hold everything the common transition needs -- suffix link, length and the
// MOST FREQUENT successor (rank 0 of the top-k order) -- and the other three words are touched only when that misses:
//   w0 = {link, length | SAMD_SINGLE (deg <= 1), e0.tok, e0.dst}
//   w1 = {aux, deg, e1.tok, e1.dst}          aux = cnt_endpos (KIND_COUNT) or min_endpos (KIND_ENDPOS)
//   w2 = {e2.tok, e2.dst, e3.tok, e3.dst}    edges 0..4 in top-k order (count desc, ties in dict order;
//   w3 = {e4.tok, e4.dst, spill, reserved}   SO/sam/static_sam.py:140-146); empty slots = (-1,-1)
// spill = index of this state's spill block (deg > 5) or -1.  Spill block of a state with deg > 5:
// [rank5, rank6, rank7 (padded (-1,-1))] followed by an open-addressing hash table of the deg-5 edges of rank >= 5:
// samd_spill_slots(deg) slots (a power of two, load factor <= 1/2), slot = samd_spill_hash(tok), linear probing,
// empty = (-1,-1).  A high-degree state (short context) is resolved in ~1.5 probes of one or two adjacent lines;
// a sorted block would cost log2(deg) DEPENDENT loads, which is what a wavefront then waits for (profiles/).
#define SAMD_SINGLE 0x40000000     // flag bit in `length`: the state has at most one outgoing edge
#define SAMD_RUN 0x20000000        // flag bit in `length`: the chain word of e0.dst holds >= 2 tokens (worth loading after following e0)
#define SAMD_LEN_MASK 0x1FFFFFFF   // the length itself
// CHAIN WORDS (device-only, derived from the node image at upload): chain[s] = the rank-0 tokens of the non-branching run that
// starts at s -- token j is e0.tok of state s + j as long as e0.dst of every state s .. s + j is the NEXT state in memory
// (the builder numbers states in creation order, which makes a corpus position's successor state the next index: 93 % of
// the states of the bench automaton) -- as 8 x u16 (vocab <= 32767: 15-bit tokens) or 4 x u32, terminator all-ones; the top bit of an
// entry is CLEAR when state s + j is SAMD_SINGLE and its suffix link is a child of the root (sam_device.h, "Round 4").  A cursor that holds
// chain[s] follows up to 8 matching tokens WITHOUT touching memory (st_transfer_chain, sam_device.h): the batched walk runs at
// the memory system's request rate, and this removes ~0.37 of its requests (scripts/walk_chain_sim.py).
struct __attribute__((aligned(64))) SamNode {
    int32_t link, length, e0_tok, e0_dst;
    int32_t aux, deg, e1_tok, e1_dst;
    int32_t e2_tok, e2_dst, e3_tok, e3_dst;
    int32_t e4_tok, e4_dst, spill, reserved;
};
static_assert(sizeof(SamNode) == 64, "SamNode must be one 64-byte line");
// int32 word index of inline edge k's token inside the node (its target follows)
#define SAMD_EDGE_WORD(k) ((k) == 0 ? 2 : 4 + 2 * (k))

struct SamEdge { int32_t tok, dst; };

#if defined(__HIPCC__)
#define SAMD_HD __host__ __device__
#else
#define SAMD_HD
#endif
// number of hash slots of a spill block (excluding the 3-entry head) for a state of degree deg > SAMD_INLINE_EDGES
SAMD_HD static inline uint32_t samd_spill_slots(int32_t deg) {
    uint32_t need = 2u * (uint32_t)(deg - SAMD_INLINE_EDGES), m = 4;
    while (m < need && m < 0x80000000u) m <<= 1;          // (a degree from a damaged image must not turn this into an endless loop)
    return m;
}
// tokens per chain word: 8 x 15-bit entries while every token id fits 15 bits (the all-ones entry is the terminator), else 4 x 31-bit; the ONE
// place this rule lives (StaticDev::chain_w, the chain-word builder, the bigram and edge table entry forms all follow it)
SAMD_HD static inline int samd_chain_w(int64_t vocab) { return vocab <= 32767 ? 8 : 4; }
// slot of the token pair (a, b) in the bigram table (StaticDev), before masking
SAMD_HD static inline uint32_t samd_bigram_hash(int32_t a, int32_t b) {
    uint32_t h = (uint32_t)a * 0x9E3779B1u + (uint32_t)b * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    return h ^ (h >> 13);
}
// slot of the edge (state, tok) in the edge table (StaticDev), before masking
SAMD_HD static inline uint32_t samd_edge_hash(int32_t state, int32_t tok) {
    uint32_t h = (uint32_t)state * 0x85EBCA77u ^ ((uint32_t)tok * 0x9E3779B1u + 0x7F4A7C15u);
    h ^= h >> 16; h *= 0x2C1B3C6Du;
    return h ^ (h >> 15);
}
SAMD_HD static inline uint32_t samd_spill_hash(int32_t tok, uint32_t slots) {
    return (((uint32_t)tok * 0x9E3779B1u) >> 7) & (slots - 1);
}
// EDGE BLOCKS (StaticDev): kinds of a fail header, the pieces of a block reference, the first slot a token probes inside a block
enum { SAMD_FK_ROOT = 0, SAMD_FK_ROOTCHILD = 1, SAMD_FK_HUB = 2, SAMD_FK_STATE = 3 };
#define SAMD_EB_IDX_MASK 0x07FFFFFFu          // 27 bits: state indices, first slots, lengths in hot words
#define SAMD_EB_HUB 0x08000000u               // slot.y: dst owns a block (slot.z = its reference)
#define SAMD_EB_KIND_SHIFT 28                 // slot.y / hot.y >> 27: kind(link)  (slot.y uses bits 28-29, hot.y bits 27-28)
#define SAMD_EB_ROOTCHILD 0x40000000u         // slot.y / bigram entry: dst is the root child of the probed token
// DISPLACED (block slots: bit 31 of y; bigram entries of a handle with blocks: bit 29 of the dst word): some key whose HOME is this slot was
// stored further along the probe sequence.  A lookup whose home slot holds another key (or nothing) and does NOT carry the bit is a
// conclusive miss after ONE probe -- without it a miss probes on to the first empty slot, a second dependent round in a quarter of the
// cases at 4 slots per entry, and a lock-step wave pays it whenever any of its lanes does (profiles/r06_walk.md)
#define SAMD_EB_DISPLACED 0x80000000u
#define SAMD_BG_DISPLACED 0x20000000u
SAMD_HD static inline uint32_t samd_eb_base(uint32_t ref) { return (ref & SAMD_EB_IDX_MASK) << 2; }     // stored in units of 4 slots (every block is a power of two >= 4)
SAMD_HD static inline uint32_t samd_eb_mask(uint32_t ref) { return (1u << (ref >> 27)) - 1u; }
SAMD_HD static inline uint32_t samd_eb_hash(int32_t tok) { uint32_t h = (uint32_t)tok * 0x9E3779B1u; return h ^ (h >> 15); }
// bits of a block slot's x word that hold the token: the smallest width whose all-ones value is not a token id
SAMD_HD static inline int samd_eb_tok_bits(int64_t vocab) { int b = 8; while (b < 31 && ((int64_t)1 << b) - 1 < vocab) b++; return b; }

// device view of a static automaton
struct StaticDev {
    const SamNode *nodes;
    const int32_t *root_next;   // dense transition table of state 0: root_next[tok] = dst or -1
    const SamEdge *spill;
    const int32_t *text;        // KIND_ENDPOS: input_ids with the -1 sentinel at [0]
    int32_t n_states, vocab, n_text, kind;
    const uint4 *chain;         // chain words (may be null: every transition then goes through the nodes)
    int32_t chain_w;            // tokens per chain word: 8 (u16) or 4 (u32)
    // BIGRAM TABLE (device-only, derived at upload; may be null; round 4: replaces round 3's per-child hashed blocks).  The states one token
    // below the root are where a walk lands after every mismatch, and in a real corpus they have the highest degrees of the automaton.  ONE
    // open-addressing table holds every edge of every root child, keyed by the token pair: (a, b) -> {state of the string "a b" ..., first
    // half of its chain word}.  A cursor that sits on the child of `a` therefore needs nothing from memory to know where to look for `b` --
    // and, since its edges are the only thing a walk ever asks of such a state, it needs no state index either: the walk kernels carry
    // "on the child of token a" as idx = -2 - a and resolve it (root16[a].x) only where an index is written out.  Landing there -- from the
    // root, or from a failed lookup (no edge = the child visited too, its suffix link is the root: its shortest string has length 1) -- costs
    // no request at all: whether token t has a child is one bit of rc_bits (vocab bits, copied to LDS by the batched walk).
    //   entry, vocab <= 32767 (chain_w 8): {a | b << 15 | lb << 30, dst, chain[dst].x, chain[dst].y}
    //   entry, larger vocabularies (chain_w 4): {a | (lb & 1) << 31, b | (lb >> 1) << 31, dst, chain[dst].x}
    //   lb = min(length[child of a] - 1, 3): what a climbing cursor's match length becomes there (3 = read root16[a].w); empty = all-ones;
    //   slot = samd_bigram_hash(a, b) & bigram_mask, linear probing, load factor <= 1/16 by default (sam_kernels.hip derive_root_hash: why).
    // root16[tok] = {dst, 0, 0, length[dst]} (dst = -1: no child).
    const uint4 *root16;
    const uint4 *bigram;
    uint32_t bigram_mask;
    const uint32_t *rc_bits;
    // EDGE TABLE (device-only, derived at upload; may be null; round 5).  The bigram table generalised to every BRANCHING state (degree >= 2,
    // any depth; the root excepted): one open-addressing table keyed by (state, token) holds every edge of every such state,
    //   entry {state, token, dst | hub(dst) << 31, first entries of chain[dst]}   (16 B; empty = all-ones; slot = samd_edge_hash & edge_mask)
    // so a transition out of a branching state is ONE probe -- hit or conclusive miss -- where the node costs word 0, then words 1-3, then a
    // spill probe (three dependent rounds for a hub).  On a natural-language-like corpus (Zipfian vocabulary: hubs of degree 10^2-10^3 at
    // depth 1-3, 16 % of the states branching) that is where a walk spends its time: every restart climbs through 3-5 short contexts, and a
    // lock-step wave pays the slowest lane's rounds at every token (profiles/r05_walk_sweep.md).  hub(dst) = dst is itself branching: the
    // cursor then probes here again next time without looking at dst's node at all.  A climb issues a hop's node word 0 (length, link, the
    // only edge of a non-branching state) and its probe together: one round per hop.  The bigram table's entries carry the same hub bit.
    const uint4 *ehash;
    uint32_t edge_mask;
    // EDGE BLOCKS + HOT WORDS (device-only, derived at upload; may be null; round 6: they REPLACE the edge table when they fit).  On a
    // natural-language-like automaton the edge-table walk still paid TWO requests per hop of a climb (the hop target's node word 0 for its
    // link and length + the probe of its edges) and a node word after every miss at a hub: 0.76-0.93 requests per visited state against
    // 0.37 on the headline corpus (profiles/r05_walk_sweep.md).  Here every branching state below the root children owns a BLOCK of 16-byte
    // slots (a power of two >= slots-per-entry x degree, at least 4; linear probing inside the block) and EVERY slot of a block -- used or
    // empty -- carries the owning state's FAIL HEADER: what transfer_state needs when the token is not there (static_sam.py:99-101):
    //   slot.x = tok | len(link) << tok_bits        tok all-ones = empty; len all-ones = "does not fit": read it from the node
    //   slot.y = dst (27 bits) | hub(dst) << 27 | kind(link) << 28 | rootchild(dst) << 30
    //   slot.z = first entries of chain[dst] (as the edge table), or the BLOCK REFERENCE of dst when hub(dst)
    //   slot.w = ref(link): by kind -- 0 ROOT: nothing; 1 ROOTCHILD: the child's token a (the hop is one bigram probe (a, tok), conclusive);
    //            2 HUB: the link's block reference (the hop is one probe of that block: hit, or its header names the next hop);
    //            3 STATE: the link's index (the hop reads hot[link]: its only edge + its own fail header)
    //   block reference = first slot / 4 (27 bits: blocks are powers of two >= 4, so every first slot is a multiple of 4 -- 2^29 slots, 8 GB)
    //                     | log2(slots) << 27   (never 0: a block has >= 4 slots)
    // so a probe -- hit or miss -- is ONE 16-byte request that also answers "where next", and a climb costs one request per hop whatever
    // the hop target is.  rootchild(dst): dst is the root child of the probed token (its edges live in the bigram table; the cursor takes
    // the unresolved form idx = -2 - tok).  Root children own no block.
    // hot[s] (16 B per state) replaces node word 0 on this path:
    //   a SINGLE state:  hot.x = ref(link(s)) as above, hot.y = len(link(s)) (27 bits) | kind(link) << 27 | SAMD_RUN | SAMD_SINGLE
    //   a branching one: hot.x = its OWN block reference (its fail header comes with whatever slot ends the probe of that block), hot.y =
    //                    kind(link) << 27 | SAMD_RUN; a root child (kind ROOT, single or branching): hot.x = its own TOKEN -- a branching
    //                    root child met by index probes the bigram table under it
    //   hot.z = e0.tok, hot.w = e0.dst: the most frequent edge (rank 0 of the top-k order; a single state's only one; -1 without edges) --
    //           met by index, a state answers its rank-0 token in one round, as node word 0 did
    // Derivation needs every length < 2^27, every state index < 2^27 and all blocks within 2^29 slots (8 GB); otherwise -- or when the
    // device cannot spare the memory, or with SAMD_EDGE_BLOCKS=0 -- the walks keep the edge table.  Results are identical by construction
    // and by test (tests/test_gpu_sam.py: traces, cursors and visited-state counts against the oracle and against the other two paths).
    const uint4 *hot;
    const uint4 *blocks;
    int32_t eb_tok_bits;        // bits of slot.x that hold the token (the rest: len(link))
    // TOP-K COUNTS (device-only, derived at upload; may be null): topk_cnt[8 s + k] = cnt_endpos of state s's rank-k successor, the
    // numerators of the best-first tree's child probabilities (static_sam.py:205-210).  Without it an expansion is two dependent round
    // trips (the parent's node for the edges, then the children's nodes for their counts); with it one.
    const int32_t *topk_cnt;
};

// ---- session (per request) -------------------------------------------------------------------
// meta[] slots
enum {
    M_NSTATES = 0, M_NEDGES, M_NTEXT, M_LAST, M_MAXLEN, M_CUR_IDX, M_CUR_LEN, M_ST_IDX, M_ST_LEN, M_ERROR,
    M_COUNT = 16
};
// dmeta[] slots (draft)
enum { D_TYPE = 0, D_N, D_NLEAVES, D_MAXDEPTH, D_IDX_DYN, D_MATCH_DYN, D_IDX_ST, D_MATCH_ST, D_COUNT = 16 };
// verdict[] slots
enum { V_BEST = 0, V_ACCEPT, V_NEXT_NODE, V_NEXT_TOKEN, V_KV_START, V_IS_TREE, V_COUNT = 8 };
// counters[] slots
// C_T_*: time the fused step kernel spent in its phases since the request began, in s_memrealtime ticks (100 MHz = 10 ns): accept
// (eval_posterior), the dynamic automaton's extension, the static cursor's transfer, lookup + draft + buffers
enum { C_STEPS = 0, C_TOKENS, C_SEQ_STEPS, C_TREE_STEPS, C_T_ACCEPT, C_T_DYN, C_T_STATIC, C_T_DRAFT, C_COUNT = 8 };

#define SAMD_HEMPTY 0xFFFFFFFFFFFFFFFFull

struct SessionDev {
    // dynamic automaton (SO/sam/dyn_sam.py:13-35): SoA state table + one open-addressing table whose
    // slots ARE the edges: key=(state<<32|tok), dst, next = next edge of the same state in dict order
    int32_t *link, *length, *minend, *head, *tail;
    uint64_t *hkey;
    int32_t *hdst, *hnext;
    int32_t *text;              // input_ids, [0] = -1 sentinel; doubles as the committed-token history
    int32_t *meta;              // M_*
    // draft + verdict block
    int32_t *tokens, *parent, *position;
    uint64_t *mask;             // [2][SAMD_MAX_DRAFT]: row i's ancestors among nodes 0..63, then (mask_hi = mask + SAMD_MAX_DRAFT) among 64..127
    uint64_t *mask_hi;
    int32_t *retrieve;          // [SAMD_MAX_DRAFT][SAMD_MAX_DRAFT]
    int32_t *dmeta;             // D_*
    int32_t *verdict;           // V_*
    int32_t *acc_tokens, *kv_index;
    int32_t *start_token, *cache_length;
    int32_t *counters;
    uint32_t hmask;
    int32_t cap_states, cap_text, max_tokens;
    // PUSHED REPORT (round 4; null until samd_session_report_target): a host-coherent block of SAMD_REPORT_INTS + 1 words the step kernel
    // writes itself when it is done -- the report, then a sequence number behind a system-scope release -- so that the host learns the
    // verdict the moment the kernel has it, without a D2H copy node and a stream synchronisation behind the cache compaction
    int32_t *h_report;
    int32_t *push_seq;          // device word: pushes so far (the sequence number survives graph replays)
};

struct samd_session {
    SessionDev dev;
    void *arena;
    size_t arena_bytes;
    int32_t max_tokens;
    int32_t *h_report_host;     // host address of dev.h_report (hipHostMalloc, coherent + mapped); owned by the session
};

// host image + device image of a static automaton
struct samd_static {
    int32_t kind;
    int64_t n_states, n_edges, n_spill, vocab, n_text;
    // host image (may be released after upload in a later round; kept for export/save)
    SamNode *h_nodes;
    int32_t *h_root;
    SamEdge *h_spill;
    int32_t *h_text;
    // device image
    SamNode *d_nodes;
    int32_t *d_root;
    SamEdge *d_spill;
    int32_t *d_text;
    int uploaded;
    int borrowed;               // device image owned by the caller (samd_static_adopt_device)
    void *d_chain;              // chain words, derived on the device from d_nodes (always owned by the handle)
    void *d_root16, *d_d1hash;  // root entries and the bigram table (StaticDev), derived with the chain words; owned by the handle
    void *d_rc_bits;            // which tokens have a root child (StaticDev); owned by the handle
    void *d_topk_cnt;           // top-k counts (StaticDev), KIND_COUNT only; owned by the handle
    int64_t n_d1hash;           // the bigram table's slots (a power of two)
    void *d_ehash;              // the edge table of the branching states (StaticDev); owned by the handle
    int64_t n_ehash;            // its slots (a power of two)
    void *d_hot, *d_blocks;     // hot words + edge blocks (StaticDev, round 6); owned by the handle; when present, d_ehash is not derived
    int64_t n_block_slots, n_block_states;
};

// sam_kernels.hip: (re)derive the chain words from the device image; called by upload / adopt and lazily by the walks
#ifdef __cplusplus
extern "C"
#endif
int samd_static_derive_chain(struct samd_static *s, void *stream);

void samd_set_error(const char *fmt, ...);
void samd_set_error_detail(const char *fmt, ...);      // kept until the next samd_set_error, which appends it

#if defined(__HIPCC__)
// max / sum over the 16 lanes (l & 15) of a DPP row -- the 16 keys of an MFMA column block that share a query row -- on the DPP network:
// xor 1 / xor 2 as quad permutes, then row_half_mirror and row_mirror (after the quad steps every lane of a quad holds the quad's value,
// so mirroring pairs quads exactly as xor 4 / xor 8 would): the SAME reduction tree, hence bit-identical sums, as four __shfl_xor steps,
// which hipcc lowers to ds_bpermute_b32 -- an LDS round trip each, 32 of them per key tile in the one computing wave's critical path
// (round 6; used by k_tree_attention, k_tree_attention_rope and k_attn_block: the compute wave of a <= 16-row step is the launch's critical path once its tile has landed).
template <int CTRL> __device__ __forceinline__ float att_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, att_dpp<0xB1>(v)); v = fmaxf(v, att_dpp<0x4E>(v));       // quad_perm [1,0,3,2], [2,3,0,1]
    v = fmaxf(v, att_dpp<0x141>(v)); return fmaxf(v, att_dpp<0x140>(v));  // row_half_mirror, row_mirror
}
__device__ __forceinline__ float row16_sum(float v) {
    v += att_dpp<0xB1>(v); v += att_dpp<0x4E>(v);
    v += att_dpp<0x141>(v); return v + att_dpp<0x140>(v);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (kernel, device), not once per
// process (a process that moves to another device would otherwise launch there without it).  `done` = one bitmask per kernel
// instantiation (function-local static at the call site); devices >= 64 set it on every call.  The mask is read and updated with atomic
// operations: host threads that drive different devices may make their first call at the same time (setting the attribute twice is
// harmless, a torn or lost update of the mask would not be a defined program).
static inline hipError_t samd_reserve_lds(const void *kernel, int bytes, unsigned long long *done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((__atomic_load_n(done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {                                   // say what the device offers (a 64 KiB-LDS part cannot run these kernels)
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess)
            samd_set_error_detail("device %d offers %d bytes of LDS per workgroup, the kernel needs %d (written for gfx950: 160 KiB)", dev, optin, bytes);
    }
    if (e == hipSuccess && dev >= 0 && dev < 64) __atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELEASE);
    return e;
}
// compute units of the CURRENT device, looked up once per device (a process may drive several GPUs; relaxed atomics: see above)
static inline int samd_device_cus() {
    static int cached[64];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 256;
    if (d >= 0 && d < 64) { const int c = __atomic_load_n(&cached[d], __ATOMIC_RELAXED); if (c > 0) return c; }
    int n = 256;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    if (d >= 0 && d < 64) __atomic_store_n(&cached[d], n, __ATOMIC_RELAXED);
    return n;
}
#endif
