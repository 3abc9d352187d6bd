/*
 * samd_hip.h -- C ABI of libsamd_hip.so: the MI355X (gfx950) draft+verify hot path of SAM-Decoding.
 *
 * Drop-in boundary.  The reference is pure Python; every entry point below
 * names the reference function it replaces (paths relative to the reference repo;
 * SO/ = samd_sam_only/, S/ = samd/).  The Python packages sam-decoding_amd/samd_sam_only and
 * sam-decoding_amd/samd bind these symbols with ctypes and re-expose the reference's classes
 * (SamdConfig, DraftModel, SamdModel, DynSAM, StaticSAM, build_sam/load_sam/dump_sam ...).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  `d_` = device pointer (HBM), `h_` = host pointer.
 *   - every function returns an int status: 0 = ok, <0 = SAMD_E_* ; never throws across the ABI.
 *   - `stream` is a hipStream_t passed as void*; all device work is enqueued on it, no internal
 *     synchronisation unless the name ends in `_sync` or the doc says "host result".
 *   - handles are not thread-safe; distinct handles are independent.
 *   - device tensors handed in (logits, KV cache, q/k/v) are BORROWED: the caller keeps them alive.
 */
#ifndef SAMD_HIP_H
#define SAMD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMD_OK 0
#define SAMD_E_INVALID (-1)   /* bad argument */
#define SAMD_E_CAPACITY (-2)  /* arena / table capacity exceeded */
#define SAMD_E_HIP (-3)       /* a HIP runtime call failed (see samd_last_error) */
#define SAMD_E_IO (-4)        /* file could not be read / written / has a bad header */
#define SAMD_E_NODEVICE (-5)  /* no gfx950 device visible */

/* A draft (sequence or tree) holds at most 128 nodes (round 5; 64 before): one wavefront builds it two nodes per lane, its ancestor mask is
 * two u64 words per node (mask / mask_hi), the verify forward runs it as two 64-row tiles.  The reference takes any max_predicts /
 * n_predicts (samd_sam_only/sam/static_sam.py:183, samd/sam/dyn_sam.py:107-113); its shipped configurations use 40-63.  The C ABI refuses
 * samd_params_t values above 128 (SAMD_E_INVALID); the Python packages above it cap larger requests at 128 with a RuntimeWarning
 * (samd_sam_only/sam/_common.py): decoding stays lossless, only the accept lengths of very long matches differ.  INTEGRATION.md section A. */
#define SAMD_MAX_DRAFT 128
#define SAMD_TOPK 8           /* SO/sam/static_sam.py:137 keeps 8 successors per state */

/* per-step report block (int32 words) copied to the host by samd_session_report_async:
 * what SamdModel.decode/update_state/generate read back per step (SO/samd_model.py:158-174, :216-235) */
#define SAMD_REP_DMETA 0      /* [16] type, n, n_leaves, max_depth, index_dyn, match_dyn, index_static, match_static, ... */
#define SAMD_REP_VERDICT 16   /* [8]  best, accept, next_node, next_token, kv_start, is_tree */
#define SAMD_REP_TOKENS 24    /* [128] candidate_tokens[best][:accept] */
#define SAMD_REP_KVINDEX 152  /* [128] retrieve[best][:accept] */
#define SAMD_REP_COUNTERS 280 /* [8]  steps, tokens, sequence steps, tree steps */
#define SAMD_REP_META 288     /* [16] n_states, n_edges, n_text, last, max_length, cursors, error flag */
#define SAMD_REPORT_INTS 304

/* automaton flavours */
#define SAMD_KIND_COUNT 0     /* samd_sam_only StaticSAM: cnt_endpos + top-k (SO/sam/static_sam.py:24-29) */
#define SAMD_KIND_ENDPOS 1    /* samd StaticSAM / both DynSAMs: min_endpos + input_ids (S/sam/static_sam.py:10-15) */

/* dtypes of borrowed floating-point tensors */
#define SAMD_F16 0
#define SAMD_BF16 1
#define SAMD_F32 2

typedef struct samd_static samd_static_t;   /* immutable corpus automaton, host image + HBM image */
typedef struct samd_session samd_session_t; /* one request stream: dynamic automaton, cursors, draft, verdict */
typedef struct samd_recycle samd_recycle_t; /* Token-Recycle [V,8] successor table */

const char *samd_last_error(void);
int samd_device_count(void);
/* make host waits on this process's device spin instead of yield (hipDeviceScheduleSpin): the decode loop waits for one
 * report per step, and the wake-up latency is ~25 us of a 3.4 ms step.  Only effective before the process creates its HIP
 * context; device < 0 = the current device.  No reference counterpart. */
int samd_host_wait_spin(int32_t device);
/* library + device facts: out[0]=ABI version, out[1]=CU count, out[2]=wavefront size, out[3]=LDS bytes/CU */
int samd_device_info(int64_t out[4]);

/* ------------------------------------------------------------------------------------------------
 * Static (corpus) suffix automaton
 * ---------------------------------------------------------------------------------------------- */

/* StaticSAM.build(batch_tokens, eos_token)  -- SO/sam/static_sam.py:31-40 (+add_batch_tokens :131-135,
 * add_state :67-96, init_topk_next :137-146); S/sam/static_sam.py:31-46 for SAMD_KIND_ENDPOS.
 * Host-side construction (sequential by nature, offline in the reference: tools/gen_sam_alpaca_sam_only.py).
 * h_tokens holds all documents back to back; document d is h_tokens[h_doc_offsets[d] .. h_doc_offsets[d+1]). */
int samd_static_build(const int32_t *h_tokens, const int64_t *h_doc_offsets, int64_t n_docs, int32_t eos_token,
                      int32_t kind, samd_static_t **out);
/* importer for automata built elsewhere (e.g. a converted reference pickle, SO/sam/utils.py:20-39):
 * per-state link/length/aux and the edges of all states, state-major, each state's edges in the
 * reference's dict (insertion) order; h_text may be NULL for SAMD_KIND_COUNT. */
int samd_static_from_tables(int32_t kind, int64_t n_states, const int32_t *h_link, const int32_t *h_length,
                            const int32_t *h_aux, const int32_t *h_deg, const int32_t *h_edge_tok,
                            const int32_t *h_edge_dst, const int32_t *h_text, int64_t n_text, samd_static_t **out);
/* load_sam of a pickle WRITTEN BY THE REFERENCE's dump_sam -- SO/sam/utils.py:20-39, S/sam/utils.py (pickle.dump of the StaticSAM object
 * graph: `states` = list of SAMState {next, link, length, cnt_endpos | min_endpos}, `input_ids` for the samd variant).  A streaming reader
 * of the pickle opcode subset that graph uses (protocols 2-5): states go straight into flat tables and on into the node image, nothing of
 * the stream is executed, peak memory ~ the image (CPython's unpickler keeps three objects per state alive: tens of GB for the published
 * 20-35 M-state automata).  kind = SAMD_KIND_COUNT (samd_sam_only) or SAMD_KIND_ENDPOS (samd).  out_params (optional, 8 doubles):
 * max_predicts, alpha, K, n_predicts, cur_index, cur_length, last, max_length as pickled (-1 = absent).  SAMD_E_IO: not such a pickle /
 * an opcode outside the subset (samd_last_error names it) -- the caller may then fall back to pickle.load. */
int samd_static_from_pickle(const char *path, int32_t kind, double out_params[8], samd_static_t **out);
/* dump_sam / load_sam  -- SO/sam/utils.py:20-39 (flat binary image instead of a pickle) */
int samd_static_save(const samd_static_t *sam, const char *path);
int samd_static_load(const char *path, samd_static_t **out);
void samd_static_free(samd_static_t *sam);
/* copy the image into HBM on the current device (idempotent); blocking */
int samd_static_upload(samd_static_t *sam);
/* out: [0]=n_states [1]=n_edges [2]=n_spill_edges [3]=vocab (root table size) [4]=device bytes [5]=kind
 *      [6]=n_text [7]=uploaded */
int samd_static_info(const samd_static_t *sam, int64_t out[8]);
/* what the upload DERIVES on the device next to the image (csrc/samd_common.h): out[0] = bytes of the chain words, out[1] = bytes of the
 * bigram table with its root entries and child bitmap, out[2] = bytes of the top-k count table, out[3] = slots of the bigram table
 * (a power of two, >= 4 x the number of root-child edges by default: SAMD_BIGRAM_SLOTS_PER_PAIR / samd_static_set_bigram_slots); zeros for
 * what was not derived; out[4] = bytes, out[5] = slots of the EDGE TABLE of the branching states (round 5: one probe per transition out of a
 * state of degree >= 2, csrc/samd_common.h; sized like the bigram table).  These bytes are resident per GPU replica NEXT to
 * samd_static_info's device bytes (the image). */
int samd_static_derived_info(const samd_static_t *sam, int64_t out[6]);
/* round 6: the EDGE BLOCKS and HOT WORDS that replace the edge table when they fit (csrc/samd_common.h: per-state blocks of 16-byte slots
 * whose every slot carries the owning state's fail header -- a probe, hit or miss, is one request that also says where transfer_state's climb
 * (static_sam.py:99-101) goes next -- and one 16-byte hot word per state in place of node word 0): out[0] = bytes of the hot words, out[1] =
 * bytes of the blocks, out[2] = their slots, out[3] = states that own a block; zeros when the handle walks through the edge table instead
 * (SAMD_EDGE_BLOCKS=0, or something did not fit: samd_static_derived_info's out[4] / out[5] are then non-zero).  Resident per GPU replica
 * next to the image, like everything samd_static_derived_info reports. */
int samd_static_edge_blocks_info(const samd_static_t *sam, int64_t out[4]);
/* re-size the bigram table of an uploaded automaton: slots_per_pair in 2 .. 64 (0 = the default, 4).  A tuning entry with no reference
 * counterpart: a sparser table only helps the BATCHED walk (samd_static_walk* / samd_static_lookup_batch: 64 cursors in lock-step pay a
 * second probe round when any collides -- 16 per pair measured best, profiles/r04_walk.md); one-cursor walks of a session are
 * indifferent.  Capped at 8 GB and at an eighth of the free device memory; results are identical at every size.  Blocking. */
int samd_static_set_bigram_slots(samd_static_t *sam, int32_t slots_per_pair, void *stream);
/* host read-back of the built automaton (tests / converters): arrays sized from samd_static_info.
 * Edges come state-major in STORED order: the first min(deg,8) are the top-k order of
 * init_topk_next (SO/sam/static_sam.py:140-146), the rest ascending by token. */
int samd_static_export(const samd_static_t *sam, int32_t *h_link, int32_t *h_length, int32_t *h_aux, int32_t *h_deg,
                       int32_t *h_edge_tok, int32_t *h_edge_dst);
/* device pointers of the HBM image, for callers that broadcast it with RCCL (multi-GPU request
 * parallelism): out[0]=nodes, out[1]=root table, out[2]=spill edges, out[3]=text; bytes in out_bytes[]. */
int samd_static_device_image(const samd_static_t *sam, void *out_ptrs[4], int64_t out_bytes[4]);
/* host pointers of the flat image (same four regions), for a host-side broadcast (gloo) or custom I/O */
int samd_static_host_image(const samd_static_t *sam, void *out_ptrs[4], int64_t out_bytes[4]);
/* rebuild a handle from the four regions received from another rank: host copy (then samd_static_upload) ... */
int samd_static_from_host_image(const int64_t info[8], const void *const h_ptrs[4], samd_static_t **out);
/* ... or adopt device buffers the caller owns (e.g. torch tensors filled by an RCCL broadcast); the handle borrows them */
int samd_static_adopt_device(const int64_t info[8], void *const d_ptrs[4], samd_static_t **out);
/* create an un-filled device image of the same shape on this rank (to receive a broadcast) */
int samd_static_alloc_like(const int64_t info[8], samd_static_t **out);

/* transfer_tokens / lookup over B independent cursors  -- SO/sam/static_sam.py:98-125.
 * d_cursors: int32 [B][2] = (index, length) per stream.  d_tokens: int32 [T][B] (time-major).
 * Each stream consumes its T tokens in order.  commit=1 writes the final cursors back
 * (transfer_tokens), commit=0 leaves them (lookup).  d_trace (optional) receives int32 [T][B][2]
 * = the (index,length) after every token.  This is the batched-streams form of the walk kernel
 * whose HBM traffic bench.py reports. */
int samd_static_walk(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                     int32_t commit, int32_t *d_trace, void *stream);

/* same launch; additionally adds the number of states visited by all streams to *d_visited (u64 on the
 * device).  bench.py multiplies it by 16 B to get the algorithmic bytes. */
int samd_static_walk_counted(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t commit, uint64_t *d_visited, void *stream);

/* StaticSAM.lookup over B cursors -- SO/sam/static_sam.py:122-125: the pair transfer_state reaches is RETURNED, cur_index / cur_length
 * stay.  d_cursors int32 [B][2] (read only), d_tokens int32 [T][B] (time-major; T = 1 is the reference's lookup, T > 1 the same over a
 * token run), d_out int32 [B][2] = the (index, length) every stream ends on.  d_visited (optional, u64 on the device) accumulates the
 * visited states.  This is the launch bench.py times for `roofline`: it produces the walk's result (8 B per stream written). */
int samd_static_lookup_batch(const samd_static_t *sam, const int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t *d_out, uint64_t *d_visited, void *stream);

/* The same walk over STREAM-MAJOR tokens: d_tokens int32 [B][T] (stream b's tokens contiguous -- what a caller that holds B token
 * sequences has), d_trace (optional) int32 [B][T][2].  Results are identical to samd_static_walk's: the token matrix is transposed
 * on the device (and the trace back), then the same kernel runs.  -- SO/sam/static_sam.py:98-125 */
int samd_static_walk_streams(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                             int32_t commit, int32_t *d_trace, void *stream);
int samd_static_walk_streams_counted(const samd_static_t *sam, int32_t *d_cursors, const int32_t *d_tokens, int32_t B, int32_t T,
                                     int32_t commit, uint64_t *d_visited, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Session = one request stream (what DraftModel + DynSAM + the StaticSAM cursor hold in the reference)
 * ---------------------------------------------------------------------------------------------- */

/* draft hyper-parameters: SamdConfig of both variants (SO/samd_config.py:9-17, S/samd_config.py:9-43) */
typedef struct samd_params {
    int32_t variant;        /* 0 = samd_sam_only rule (SO/draft.py:50-59), 1 = samd rule (S/draft.py:52-63) */
    int32_t max_predicts;   /* SO: cap of the draft size */
    double alpha;           /* SO: n = min(max_predicts, 1 + int(match * alpha)) */
    int32_t K;              /* SO: per-depth width of the tree search */
    int32_t len_bias;       /* both */
    int32_t n_predicts;     /* S: fixed sequence-draft length */
    int32_t len_threshold;  /* S: SAM draft iff max(match) >= threshold */
    int32_t static_null;    /* S: 1 = NullStaticSAM (S/sam/static_sam.py:128-137) */
    int32_t reserved;
} samd_params_t;

/* host view of a finished draft (returned by samd_session_read_draft; host result => synchronises) */
typedef struct samd_draft_host {
    int32_t type;                           /* 0 = sequence, 1 = tree, 2 = defer to the tree model (S variant) */
    int32_t n;                              /* number of draft nodes */
    int32_t n_leaves, max_depth;            /* shape of retrieve */
    int32_t index_dyn, match_dyn, index_static, match_static;   /* lookup results (match_static after len_bias) */
    int32_t tokens[SAMD_MAX_DRAFT];
    int32_t parent[SAMD_MAX_DRAFT];         /* anc_tree; sequence: i-1 */
    int32_t position[SAMD_MAX_DRAFT];       /* tree_position_ids / seq_position_ids */
    uint64_t mask[SAMD_MAX_DRAFT];          /* row i: bit j set iff node i attends node j < 64 (tree_attn_mask) */
    uint64_t mask_hi[SAMD_MAX_DRAFT];       /* row i: bit j set iff node i attends node 64 + j */
    int32_t retrieve[SAMD_MAX_DRAFT * SAMD_MAX_DRAFT];  /* [n_leaves][max_depth], -1 padded */
} samd_draft_host_t;

/* host view of a verdict (eval_posterior + update_state outputs) */
typedef struct samd_verdict_host {
    int32_t best, accept, next_node, next_token;
    int32_t tokens[SAMD_MAX_DRAFT];         /* candidate_tokens[best][:accept] */
    int32_t kv_index[SAMD_MAX_DRAFT];       /* retrieve[best][:accept] (tree) or 0..accept-1 (sequence) */
} samd_verdict_host_t;

/* DynSAM() + cursors.  max_tokens bounds prompt + generated tokens (the reference's max_cache_len). */
int samd_session_create(int32_t max_tokens, samd_session_t **out);
void samd_session_free(samd_session_t *s);
/* DraftModel.reset  -- SO/draft.py:45-47: DynSAM.reset (SO/sam/dyn_sam.py:37-43) + static cursor to root */
int samd_session_reset(samd_session_t *s, void *stream);

/* DynSAM.add_tokens  -- SO/sam/dyn_sam.py:101-105 (cursor transfer first, then add_state :50-76).
 * d_tokens: int32 [n]; if d_n != NULL the count is read from device memory (sync-free chaining). */
int samd_dyn_add_tokens(samd_session_t *s, const int32_t *d_tokens, int32_t n, const int32_t *d_n, void *stream);
/* DynSAM.transfer_tokens (commit=1) / lookup (commit=0, n=1)  -- SO/sam/dyn_sam.py:78-114.
 * d_out (optional): int32 [2] final (index,length). */
int samd_dyn_walk(samd_session_t *s, const int32_t *d_tokens, int32_t n, int32_t commit, int32_t *d_out, void *stream);
/* StaticSAM.transfer_tokens / lookup on the session's static cursor (single stream) */
int samd_session_static_walk(samd_session_t *s, const samd_static_t *sam, const int32_t *d_tokens, int32_t n,
                             const int32_t *d_n, int32_t commit, int32_t *d_out, void *stream);
/* host read-back of the dynamic automaton (tests): out_info: [0]=n_states [1]=n_edges [2]=n_text [3]=last
 * [4]=max_length [5]=cur_index [6]=cur_length [7]=static cur_index [8]=static cur_length [9]=error flag.
 * Pass NULL arrays to query sizes only.  Edges state-major in dict (insertion) order.  Synchronises. */
int samd_session_export(samd_session_t *s, int64_t out_info[10], int32_t *h_link, int32_t *h_length, int32_t *h_minend,
                        int32_t *h_deg, int32_t *h_edge_tok, int32_t *h_edge_dst, int32_t *h_text, void *stream);
/* set the cursors explicitly (tests; mirrors assigning cur_index/cur_length in the reference objects) */
int samd_session_set_cursors(samd_session_t *s, int32_t dyn_index, int32_t dyn_length, int32_t st_index,
                             int32_t st_length, void *stream);

/* seed the first lookup: start_token <- *d_src (the arg-max of the last prompt position, SO/samd_model.py:110
 * feeding SO/utils.py:86); device-to-device, no host round trip */
int samd_session_set_start_token(samd_session_t *s, const int32_t *d_src, void *stream);

/* DraftModel.lookup  -- SO/draft.py:50-59 (variant 0) / S/draft.py:52-63 (variant 1):
 * both lookups, len_bias, the dyn-vs-static rule, then
 *   sequence: DynSAM.gen_draft (SO/sam/dyn_sam.py:116-121; S/sam/dyn_sam.py:99-113 with to_anc; S/sam/static_sam.py:119-125)
 *   tree:     StaticSAM.gen_draft best-first search (SO/sam/static_sam.py:182-215, CPython heapq order, float64 prob)
 * and gen_buffers (SO/sam/static_sam.py:148-180).  d_start_token: int32 [1] on the device.
 * The draft stays in the session (device); read it with samd_session_read_draft or feed it to the
 * verify kernels through samd_session_device_draft. `sam` may be NULL (empty static automaton). */
int samd_session_draft(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p,
                       const int32_t *d_start_token, void *stream);
/* forced forms of the two SO draft generators (granular API of DynSAM.gen_draft / StaticSAM.gen_draft) */
int samd_session_draft_seq(samd_session_t *s, const samd_params_t *p, int32_t index, int32_t match, int32_t start_token,
                           void *stream);
int samd_session_draft_tree(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, int32_t index,
                            int32_t match, int32_t start_token, void *stream);
/* S-variant fixed-length drafts (granular): source 0 = DynSAM (to_anc), 1 = StaticSAM */
int samd_session_draft_fixed(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, int32_t source,
                             int32_t index, int32_t start_token, void *stream);
/* install an externally produced draft (Token-Recycle / EAGLE tree) so that verify + accept can run on it:
 * tokens and parent array on the device, n on the host. */
int samd_session_set_draft(samd_session_t *s, const int32_t *d_tokens, const int32_t *d_parent, int32_t n, int32_t type,
                           void *stream);
int samd_session_read_draft(samd_session_t *s, samd_draft_host_t *out, void *stream);
/* like samd_session_set_draft, but takes effect only when the session's last lookup deferred to the tree
 * model (draft type 2: S/draft.py:63) -- lets the Token-Recycle / EAGLE draft be installed without a host
 * round trip.  reverse_leaves as in samd_tree_buffers. */
int samd_session_set_draft_if_deferred(samd_session_t *s, const int32_t *d_tokens, const int32_t *d_parent, int32_t n,
                                       int32_t reverse_leaves, void *stream);
/* SEAM EXPERIMENT HOOKS (round 4; scripts/seam_probe.py, profiles/r04_attention.md section 5 -- not used by the product path).
 * samd_tree_attention_signal = samd_tree_attention whose merge launch adds 1 to *d_arrive per (row, head) workgroup once its output is
 * visible device-wide; samd_gemm_cs_residual_early = samd_gemm_cs_residual for o_proj (K = 4096, <= 8 rows) that requests its weights at
 * entry and polls d_counter until (its own epoch + 1) * arrivals producers have arrived before it reads A -- so that it can be launched
 * on a second stream BESIDE attention.  d_epoch int32[N / 16], zero-initialised. */
int samd_tree_attention_signal(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t n_q_pad,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, const uint64_t *d_mask,
                               const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace, int64_t workspace_bytes,
                               int32_t *d_arrive, void *stream);
int samd_gemm_cs_residual_early(const void *d_A, const void *d_Wg, int32_t N, int32_t K, void *d_x, float *d_ssq, int32_t dtype, const int32_t *d_counter,
                                int32_t *d_epoch, int32_t arrivals, void *stream);

/* PUSHED REPORT (round 4).  samd_session_report_target allocates (once) a host-coherent block of SAMD_REPORT_INTS + 1 int32 for the
 * session and returns its host address; from then on samd_session_step's kernel writes the report block there itself when it is done --
 * the words, then, behind a system-scope release, a sequence number at [SAMD_REPORT_INTS] that grows by one per step.
 * samd_report_wait spins on that word until it differs from last_seq (SAMD_OK) or timeout_us has passed (SAMD_E_CAPACITY).  The host
 * thereby learns the verdict without the D2H copy node and the stream synchronisation behind the cache compaction -- what the
 * reference pays as .item() / .tolist() syncs (samd_sam_only/samd_model.py:158-174).  The block belongs to the session. */
int samd_session_report_target(samd_session_t *s, int32_t **out_host);
int samd_report_wait(const int32_t *h_report, int32_t last_seq, int64_t timeout_us);

/* enqueue one D2H copy of the SAMD_REPORT_INTS-word report block into h_dst (pinned host memory for a
 * truly asynchronous copy); no synchronisation -- the caller waits on the stream / an event. */
int samd_session_report_async(samd_session_t *s, int32_t *h_dst, void *stream);

/* device view of the session's draft/verdict block (fixed layout, see samd_session_block_t in DESIGN.md):
 * out[0]=tokens int32[128] out[1]=parent int32[128] out[2]=position int32[128] out[3]=mask u64[2][128] (low words of all rows, then the
 * high words: what samd_tree_attention reads for n_q_pad > 64) out[4]=retrieve int32[128*128] out[5]=meta int32[16] {type,n,n_leaves,max_depth,...}
 * out[6]=verdict int32[8] {best,accept,next_node,next_token,...} out[7]=accepted tokens int32[128]
 * out[8]=kv_index int32[128] out[9]=start_token int32[1] out[10]=cache_length int32[1]
 * out[11]=history int32[max_tokens] (all committed tokens) out[12]=counters int32[8] */
int samd_session_device_views(samd_session_t *s, void *out[16]);

/* gen_buffers(anc_tree) standalone  -- SO/sam/static_sam.py:148-180.  d_parent int32[n] ->
 * d_position int32[n], d_mask u64[n] (n > 64: u64[2 n], the n high words -- nodes 64..127 -- behind the n low words),
 * d_mask_bool u8[n*n] (optional), d_retrieve int32[n*n] (optional),
 * d_shape int32[2] = {n_leaves, max_depth}. reverse_leaves=1 gives Token-Recycle's row order
 * (S/tree_model/token_recycle/utils.py:88). */
int samd_tree_buffers(const int32_t *d_parent, int32_t n, int32_t reverse_leaves, int32_t *d_position, uint64_t *d_mask,
                      uint8_t *d_mask_bool, int32_t *d_retrieve, int32_t *d_shape, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Verify side
 * ---------------------------------------------------------------------------------------------- */

/* torch.argmax(logits, -1) per row, first maximum wins  -- SO/utils.py:86 and :131.
 * d_logits [rows][row_stride] of `dtype`; d_out int32 [rows]. d_rows (optional) = device row count. */
int samd_argmax_rows(const void *d_logits, int32_t dtype, int32_t rows, int64_t vocab, int64_t row_stride,
                     const int32_t *d_rows, int32_t *d_out, void *stream);

/* eval_posterior (greedy) + the token/index selection of update_state
 * -- SO/utils.py:127-141, SO/samd_model.py:158-170 -- on the session's current draft, from per-node
 * arg-max tokens d_node_argmax int32[n].  Mirrors the reference's -1 padding quirk. */
int samd_session_accept(samd_session_t *s, const int32_t *d_node_argmax, void *stream);
/* gen_candidates' gather (SO/utils.py:92-96) for the draft the session holds: d_candidates int64 [n_leaves * max_depth] =
 * (tokens + [0])[retrieve] (pad token 0 for -1 entries), d_rowmap int32 [n_leaves * max_depth] = the retrieve table; `capacity` =
 * elements either buffer can take.  Device to device, no host round trip (the shapes are in the previous step's report). */
int samd_session_candidates(samd_session_t *s, int64_t *d_candidates, int32_t *d_rowmap, int32_t capacity, void *stream);
/* samd_session_step for a verdict decided elsewhere -- the sampling branch of eval_posterior (SO/utils.py:142-184, samd_posterior_sampled*):
 * d_best_accept int32[2] = {best candidate row, accept length incl. the root}, d_next_token = the start token drawn for the next step
 * (torch.multinomial over sample_p, SO/utils.py:84); then update_state's selection (SO/samd_model.py:165-169), DraftModel.update and the
 * next lookup as in samd_session_step.  All three operands are read on the device: the step needs no host round trip before it. */
int samd_session_step_given(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p, const int32_t *d_best_accept,
                            const int32_t *d_next_token, void *stream);
int samd_session_read_verdict(samd_session_t *s, samd_verdict_host_t *out, void *stream);
/* DraftModel.update(tokens) with the accepted tokens of the last verdict  -- SO/draft.py:62-67 */
int samd_session_commit(samd_session_t *s, const samd_static_t *sam, void *stream);
/* one fused single-wavefront launch: accept -> commit -> lookup(next start token) -> draft -> buffers.
 * This is what SamdModel.decode runs between two LM forwards (SO/samd_model.py:116-156). */
int samd_session_step(samd_session_t *s, const samd_static_t *sam, const samd_params_t *p,
                      const int32_t *d_node_argmax, void *stream);

/* SamdStaticCache.select_indices  -- SO/cache.py:118-133: for every K and V tensor copy rows
 * start+idx[j] -> start+j (j < accept), then cache_length += accept.  d_tensors: device array of
 * n_tensors base pointers, each [n_heads][max_len][head_dim] of `elem_bytes`-byte elements.
 * start / idx / accept are read from the session (device side); is_tree==0 (sequence) skips the copy. */
/* eval_posterior's sampling branch on the device (SO/utils.py:142-184).  d_probs [n_candidates * depth][vocab] (f16 / bf16 / f32) =
 * softmax of the WARPED logits of every (candidate row, position) -- the caller applies SamdGenerationConfig.logits_processor to the
 * whole batch on the device; d_candidates int64 [n_candidates][depth] (-1 padded); d_uniforms double [n_uniforms] = the next values
 * of the host's random.random() stream, in order: THE RNG CONTRACT -- the k-th uniform examined in a step is the k-th value the
 * reference would have drawn (one per distinct token tried, utils.py:165); d_out[2] says how many were consumed, and the caller
 * restores its generator and advances it by exactly that count.  d_work [vocab] (dtype of d_probs) = scratch; when d_out[3] == 1 it
 * holds the renormalised residual distribution the reference returns as sample_p (utils.py:176-177), otherwise sample_p is the
 * softmax of the RAW logits of (row d_out[0], position d_out[1] - 1).  d_out int32[5] = {best row, accept length, uniforms consumed,
 * residual flag, status (1 = ran out of uniforms)}.  Rounding follows torch on tensors of the same dtype. */
int samd_posterior_sampled(const void *d_probs, int32_t dtype, const int64_t *d_candidates, int32_t n_candidates, int32_t depth, int64_t vocab,
                           const double *d_uniforms, int32_t n_uniforms, void *d_work, int32_t *d_out, void *stream);
/* the same with ONE probability row per draft node instead of one per (candidate, position) cell: d_probs [n_rows][vocab], cell
 * (row, position) reads node d_rowmap[row * depth + position] (the retrieve table, SO/samd_model.py:144; -1 = the last node).  The
 * warpers then run over <= 64 rows instead of leaves x depth (a 60-leaf tree of depth 10: 600 top-p sorts before). */
int samd_posterior_sampled_nodes(const void *d_probs, int32_t dtype, const int32_t *d_rowmap, int32_t n_rows, const int64_t *d_candidates,
                                 int32_t n_candidates, int32_t depth, int64_t vocab, const double *d_uniforms, int32_t n_uniforms,
                                 void *d_work, int32_t *d_out, void *stream);
int samd_kv_compact(samd_session_t *s, void *const *d_tensors, int32_t n_tensors, int32_t n_heads, int64_t max_len,
                    int32_t head_dim, int32_t elem_bytes, void *stream);

/* the same compaction with host-known start/accept and a device index vector: the stand-alone form of
 * SamdStaticCache.select_indices(indices, accept_length) -- SO/cache.py:118-133 */
int samd_kv_compact_indices(void *const *d_tensors, int32_t n_tensors, int32_t n_heads, int64_t max_len, int32_t head_dim,
                            int32_t elem_bytes, int32_t start, const int32_t *d_indices, int32_t accept, void *stream);

/* SamdStaticCache.reset / set_length  -- SO/cache.py:89-94: the committed KV length lives in the session
 * (device scalar) so that accept, compaction and attention chain without host round trips. */
int samd_session_set_cache_length(samd_session_t *s, int32_t length, void *stream);
int samd_session_get_cache_length(samd_session_t *s, int32_t *h_out, void *stream);   /* host result */

/* L2 warm-up hint: the projection that FOLLOWS a glue launch on the stream.  The glue launch then carries one extra workgroup per
 * workgroup of that samd_gemm_skinny launch, which reads the first kb_per_workgroup KiB of its weight stream (packed layout) into
 * the L2 of the XCD that will run it -- HBM idles during the glue, and what a kernel reads stays in the XCD L2s for the next
 * kernel (csrc/warm_device.h, scripts/probes/l2_retain_probe.hip).  Purely a speed hint: results never depend on it.
 * No reference counterpart (the reference's forward is HF's; call sites SO/samd_model.py:134-138). */
typedef struct samd_warm {
    const void *d_packed_w;        /* samd_gemm_pack_weights output of the next projection; NULL = no warm-up */
    int32_t N, K, splits;          /* its shape and split-K factor (samd_gemm_splits; 1 for samd_gemm_skinny_silu / lm_head) */
    int32_t kb_per_workgroup;      /* KiB to warm per projection workgroup (<= 0: none) */
    int32_t delay;                 /* the warm workgroups first sleep delay x 64 cycles, so that the glue's own loads queue first */
    int32_t where;                 /* samd_tree_attention_warm: 0 = warm from the split launch, 1 = from the merge launch */
} samd_warm_t;

/* tree-mask attention of the n draft tokens over L cached + n new keys
 * -- SO/model_patch/llama.py:82-96 (mask semantics) + the SDPA call it feeds.
 * q [n_q_pad][H][D], k_cache/v_cache [H_kv][max_len][D] (new rows already written at [L, L+n)),
 * out [n_q_pad][H][D] (rows >= n are zeroed); mask u64[n_q_pad] row i bit j = node i attends node j; n_q_pad <= 128 (round 5), run as
 * 64-row tiles: for n_q_pad > 64 the mask is u64[2][SAMD_MAX_DRAFT] -- the low words (nodes 0..63) of all rows, then, SAMD_MAX_DRAFT
 * entries further, the high words (nodes 64..127).  L and n are read from device
 * memory (d_cache_length, d_n).  head_dim must be 128, dtype f16 or bf16.  d_workspace: scratch of
 * samd_tree_attention_workspace() bytes (split-KV partials). */
int64_t samd_tree_attention_workspace(int32_t n_q_pad, int32_t n_heads, int32_t head_dim);
int samd_tree_attention(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype,
                        int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                        const uint64_t *d_mask, const int32_t *d_cache_length, const int32_t *d_n, float scale,
                        void *d_workspace, int64_t workspace_bytes, void *stream);
/* round 6: the same over a cache whose V half is kept TRANSPOSED, d_vt_cache [H_kv][D][max_len] (the layout samd_attention_block reads);
 * max_len a multiple of 8.  Same sums in the same order as samd_tree_attention: bit-identical output.  At n_q_pad <= 16 one wave per (head,
 * KV split) reads its V^T operands straight from memory (no LDS staging, no workgroup barrier): 11.9 -> 11.0 us per layer at 8 rows over 800
 * cached keys (profiles/r06_attention.md).  `next` as in samd_tree_attention_warm (may be NULL).  The cache is maintained by
 * samd_gemm_qkv_rope_vt / samd_gemm_qkv_rope_norm_vt / samd_rope_kv_write_vt / samd_rope_kv_write_cs_vt (new rows), samd_kv_compact*_vt (accepted
 * rows) and read by samd_prefill_attention_vt (the prompt). */
int samd_tree_attention_vt(const void *d_q, const void *d_k_cache, const void *d_vt_cache, void *d_out, int32_t dtype,
                           int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                           const uint64_t *d_mask, const int32_t *d_cache_length, const int32_t *d_n, float scale,
                           void *d_workspace, int64_t workspace_bytes, const samd_warm_t *next, void *stream);
/* the same; its merge launch also warms the L2 for the output projection that follows (next may be NULL) */
int samd_tree_attention_warm(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype,
                             int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                             const uint64_t *d_mask, const int32_t *d_cache_length, const int32_t *d_n, float scale,
                             void *d_workspace, int64_t workspace_bytes, const samd_warm_t *next, void *stream);

/* The attention block of one decoder layer in ONE launch (csrc/attn_kernels.hip): RoPE on q and k, the K/V row write of
 * SamdStaticCache.update (SO/cache.py:103-115) at [write_pos, write_pos + n), tree-mask attention (SO/model_patch/llama.py:82-96 +
 * SDPA) and the merge of the per-tile partial softmaxes -- what samd_rope_kv_write + samd_tree_attention do in three launches.
 *   d_qkv: the q|k|v projection's output [rows >= n_q_pad][(H + 2 H_kv) * D] of `dtype`, or (n_partials > 0) the streaming
 *          GEMM's fp32 partial sums [n_partials][partial_stride] of the same layout (summed, then rounded to dtype);
 *   d_cs:  float [n_q_pad][D]: cos (first D/2) and sin (last D/2) of every row's position -- samd_rope_rows, once per forward
 *          (position of row r = *d_base + d_rel_pos[r], clamped to the table);
 *   d_k_cache [H_kv][max_len][D];  d_vt_cache [H_kv][D][max_len] -- V is cached TRANSPOSED (samd_rope_kv_write_vt and
 *          samd_kv_compact*_vt write / compact that layout); max_len must be a multiple of 8;
 *   visibility: keys < *d_visible_len are visible to every row; key *d_visible_len + j is visible to row i iff bit j of
 *          d_mask[i] is set.  d_visible_len == NULL means *d_write_pos (the base model's verify: bit j = new row j).  A draft
 *          head's tree level keeps the rows of earlier levels in its cache: visible_len = accepted length, write_pos beyond it.
 * out [n_q_pad][H][D], rows >= n zeroed.  head_dim 128, f16 / bf16, n_q_pad <= 64. */
int samd_rope_rows(const int32_t *d_rel_pos, const int32_t *d_base, const float *d_cos, const float *d_sin, float *d_cs, int32_t rows,
                   int32_t head_dim, int32_t max_pos, void *stream);
int samd_attention_block(const void *d_qkv, int32_t n_partials, int64_t partial_stride, const float *d_cs, void *d_k_cache, void *d_vt_cache,
                         void *d_out, int32_t dtype, int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                         const uint64_t *d_mask, const int32_t *d_write_pos, const int32_t *d_visible_len, const int32_t *d_n, float scale,
                         void *stream);
/* samd_rope_kv_write with the rows' cos | sin taken from d_cs (float [rows][128], samd_rope_rows -- once per forward) instead of the
 * position tables: one memory round trip instead of two (the table lookup depends on L and the row's position). */
int samd_rope_kv_write_cs(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                          const float *d_cs, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t rows,
                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype,
                          int32_t n_partials, int64_t partial_stride, void *stream);
/* round 6: the same writing V into a TRANSPOSED cache, d_vt_cache [H_kv][D][max_len] (samd_tree_attention_vt) */
int samd_rope_kv_write_cs_vt(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                             const float *d_cs, void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t rows,
                             int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype,
                             int32_t n_partials, int64_t partial_stride, void *stream);
/* samd_rope_kv_write + samd_tree_attention in TWO launches instead of three (row-major V cache, csrc/verify_kernels.hip
 * k_tree_attention_rope): the 16 KV splits rotate their Q rows themselves, one more workgroup per head owns the n new keys (rotates
 * k, writes the K / V rows at [L, L + n), attends to them), k_attn_combine_slots merges the 17 slots.  d_qkv / n_partials /
 * partial_stride / d_cs as for samd_attention_block; mask bit j of row i = new key j (no visible-prefix form). */
int64_t samd_tree_attention_rope_workspace(int32_t n_q_pad, int32_t n_heads, int32_t head_dim);
int samd_tree_attention_rope(const void *d_qkv, int32_t n_partials, int64_t partial_stride, const float *d_cs, void *d_k_cache, void *d_v_cache,
                             void *d_out, int32_t dtype, int32_t n_q_pad, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len,
                             const uint64_t *d_mask, const int32_t *d_cache_length, const int32_t *d_n, float scale, void *d_workspace,
                             int64_t workspace_bytes, void *stream);
/* samd_rope_kv_write with the V cache transposed ([H_kv][D][max_len]); d_vt_cache may be NULL (q and K only).
 * samd_kv_compact / samd_kv_compact_indices over a pointer table whose LAST n_transposed tensors are transposed (2-byte elements):
 * select_indices (SO/cache.py:118-133) for the K tensors followed by the V^T tensors. */
int samd_rope_kv_write_vt(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                          const float *d_cos, const float *d_sin, void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t rows,
                          int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t max_pos, int32_t dtype,
                          int32_t n_partials, int64_t partial_stride, void *stream);
int samd_kv_compact_vt(samd_session_t *s, void *const *d_tensors, int32_t n_tensors, int32_t n_transposed, int32_t n_heads, int64_t max_len,
                       int32_t head_dim, int32_t elem_bytes, void *stream);
int samd_kv_compact_indices_vt(void *const *d_tensors, int32_t n_tensors, int32_t n_transposed, int32_t n_heads, int64_t max_len, int32_t head_dim,
                               int32_t elem_bytes, int32_t start, const int32_t *d_indices, int32_t accept, void *stream);

/* ---- EAGLE-2's tree logic between two forwards of the draft head (csrc/eagle_kernels.hip; reference: Eagle2Model.topk_genrate,
 * S/tree_model/eagle2/eagle2_model.py:848-913 the level loop, :893-913 the re-rank).  samd_e2_state_t holds device pointers to small
 * caller-allocated arrays (k = 8 = top_k, depth <= 7):
 *   row_lse f32[8], top_logp f32[8][8], top_idx i32[8][8]     -- samd_e2_rowstats: per row log-sum-exp and top-k (log-prob, token)
 *   scores f32[16] (two level parities), cs_index i32[8], mask_rows u64[64], row_src i32[8], ids i32[8]   -- the current level's rows (samd_e2_select)
 *   all_scores f32[8 + 64 depth], all_tokens i32[same], parents_list i32[1 + 8 depth]   -- the reference's scores_list / ss_token / parents_list
 *   rec_top_vals/idx [1 + depth][8][8], rec_best_vals/idx [depth][8], rec_final_vals/idx [keep]   -- every top-k decision, in the reference's order
 * samd_e2_select(level = -1) installs the root's 8 candidates as level 0's rows; level >= 0 picks the 8 rows of level + 1 from the
 * 64 cumulative scores; both stage the fc projection's input rows [embed[token] | parent hidden state] into d_fc_in [8][2 hidden] and
 * the rows' ancestor masks (bit j = tree row j) into mask_rows.  samd_e2_finish keeps the best `keep` candidates in candidate order
 * -> tokens / parents [keep + 1] (parent -1 for the root).  samd_sum_partials_bias: out = sum of fp32 partials + bias, rounded. */
typedef struct samd_e2_state {
    float *row_lse; float *top_logp; int32_t *top_idx; float *scores; int32_t *cs_index; float *all_scores; int32_t *all_tokens;
    int32_t *parents_list; uint64_t *mask_rows; int32_t *row_src; int32_t *ids;
    float *rec_top_vals; int32_t *rec_top_idx; float *rec_best_vals; int32_t *rec_best_idx; float *rec_final_vals; int32_t *rec_final_idx;
} samd_e2_state_t;
/* d_workspace (samd_e2_rowstats_workspace(vocab) bytes; 0 = vocabulary too large for the split form): every row is spread over
 * vocab / 4096 workgroups + one merge launch (f16 / bf16 logits); NULL: one workgroup per row (also the f32 form). Same results. */
int64_t samd_e2_rowstats_workspace(int64_t vocab);
int samd_e2_rowstats(const void *d_logits, int32_t dtype, int32_t rows, int64_t vocab, int64_t row_stride, const samd_e2_state_t *st, void *d_workspace,
                     int64_t workspace_bytes, void *stream);
int samd_e2_select(const samd_e2_state_t *st, int32_t level, const void *d_hidden, const void *d_embed, int32_t hidden, int32_t vocab, void *d_fc_in,
                   int32_t *d_rel_pos /* optional: [8] <- level + 1 */, int32_t *d_L, int32_t *d_Lw /* optional pair: L += advance_L (the accepted
                   tokens the extension forward just wrote), Lw <- L + 8 (level + 1) = the cache row of the new level */, int32_t *d_n /* optional: <- 8 */,
                   int32_t advance_L, int32_t dtype, void *stream);
/* the accepted tokens of one verified step -> the draft head's extension forward, device to device (reference: the plugin's
 * update() + the first forward of topk_genrate, S/tree_model/eagle2/eagle2.py:37-63, eagle2_model.py:835-846): row t < n_accepted of
 * d_fc_in [n_accepted][2 hidden] <- [embed[acc_tokens[t + 1] or, for the last row, start_token[0]] | d_hidden_rows[kv_index[t]]]
 * (kv_index -1 = row n_rows - 1, the reference's padding entry); d_rel_pos[t] = t, d_mask_rows[t] = causal chain, d_n[0] = n_accepted,
 * d_sample_token[0] (optional) = start_token[0] widened for samd_e2_finish.
 * d_kv_index / d_acc_tokens / d_start_token: the session's report block (samd_session_device_views). */
int samd_e2_stage_extend(const void *d_hidden_rows, const int32_t *d_kv_index, const int32_t *d_acc_tokens, const int32_t *d_start_token, int32_t n_accepted,
                         int32_t n_rows, const void *d_embed, int32_t hidden, int32_t vocab, void *d_fc_in, int32_t *d_rel_pos, uint64_t *d_mask_rows, int32_t *d_n,
                         int64_t *d_sample_token, int32_t dtype, void *stream);
int samd_e2_finish(const samd_e2_state_t *st, int32_t depth, int32_t keep, const int64_t *d_sample_token, int32_t *d_tokens, int32_t *d_parents, void *stream);
int samd_sum_partials_bias(const float *d_part, int32_t n_partials, int64_t partial_stride, const void *d_bias, void *d_out, int32_t rows, int32_t N,
                           int32_t dtype, void *stream);

/* ---- memory-bound glue of the verify forward (between the library GEMMs).  The arithmetic of the
 * forward lives in HuggingFace transformers in the reference (third party, not vendored; call sites
 * SO/samd_model.py:102-106 and :134-138); these follow LlamaDecoderLayer's operators and take every
 * dynamic scalar from device memory so that one decode step is graph-capturable. ---- */
/* In the three consumers below the GEMM-output operand (delta / qkv / gate_up) may be given as fp32 split-K partial
 * sums of samd_gemm_skinny: n_partials > 0, the pointer then addresses float [n_partials][...] with `partial_stride`
 * elements between splits; the sum is rounded to `dtype` before use.  n_partials == 0: a plain `dtype` tensor. */
/* hidden = embed_tokens(input_ids) */
int samd_embed_rows(const int32_t *d_tokens, const void *d_table, void *d_out, int32_t rows, int32_t hidden, int32_t vocab,
                    int32_t dtype, void *stream);
/* LlamaRMSNorm; if d_delta != NULL first x += delta (residual add) and store x back */
int samd_rmsnorm(void *d_x, const void *d_delta, const void *d_weight, void *d_out, int32_t rows, int32_t hidden, float eps,
                 int32_t dtype, int32_t n_partials, int64_t partial_stride, void *stream);
/* the same, plus the warm-up workgroups for the projection that consumes d_out (next may be NULL) */
int samd_rmsnorm_warm(void *d_x, const void *d_delta, const void *d_weight, void *d_out, int32_t rows, int32_t hidden, float eps,
                      int32_t dtype, int32_t n_partials, int64_t partial_stride, const samd_warm_t *next, void *stream);
/* rotary embedding of q,k at positions L + rel_pos[r] (SO/samd_model.py:127-132) and
 * SamdStaticCache.update (SO/cache.py:103-115): K/V rows written at [L, L+n).  d_qkv [rows][(H+2Hkv)*D]. */
int samd_rope_kv_write(const void *d_qkv, const int32_t *d_rel_pos, const int32_t *d_cache_length, const int32_t *d_n,
                       const float *d_cos, const float *d_sin, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t rows,
                       int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t max_pos, int32_t dtype,
                       int32_t n_partials, int64_t partial_stride, void *stream);
/* Causal self-attention of the PROMPT's rows (round 5) -- what the reference runs through HF's LlamaAttention.forward / SDPA on the prompt
 * (SO/samd_model.py:102-106: the prefill call of the patched forward, attention_mask None -> causal).  q [rows][n_heads][128] (after RoPE),
 * k_cache / v_cache [n_kv_heads][max_len][128] with the prompt's K / V rows already written at [pos0, pos0 + rows) (samd_rope_kv_write) and the
 * earlier context at [0, pos0); query row i attends keys 0 .. pos0 + i.  out [rows][n_heads * 128] in the model dtype (the input of o_proj).
 * Cache rows at or beyond pos0 + rows are never used (masked / zeroed), whatever they hold.  fp32 softmax, P rounded to the model dtype for
 * the PV product (as fused SDPA kernels do).  head_dim must be 128, pos0 + rows <= max_len.  One launch, no workspace. */
int samd_prefill_attention(const void *d_q, const void *d_k_cache, const void *d_v_cache, void *d_out, int32_t dtype, int32_t rows, int32_t pos0,
                           int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, float scale, void *stream);
/* round 6: the same over a transposed V cache, d_vt_cache [n_kv_heads][128][max_len] (max_len a multiple of 8 below 2^24): the V^T tile is copied into
 * LDS (one 16-byte load + one 16-byte LDS store per 8 keys of a column) instead of transposed on the way */
int samd_prefill_attention_vt(const void *d_q, const void *d_k_cache, const void *d_vt_cache, void *d_out, int32_t dtype, int32_t rows, int32_t pos0,
                              int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, float scale, void *stream);
/* LlamaMLP activation: silu(gate) * up with gate|up concatenated per row */
int samd_silu_mul(const void *d_gate_up, void *d_out, int32_t rows, int32_t inter, int32_t dtype, int32_t n_partials,
                  int64_t partial_stride, void *stream);

/* weight-streaming skinny GEMM of the verify forward: out[m][n] = sum_k A[m][k] * W[n][k], m < rows_pad in {16,32,48,64},
 * W = an nn.Linear weight [N][K] (what the reference runs through HF's q/k/v/o/gate/up/down/lm_head projections,
 * call sites SO/samd_model.py:102-106, :134-138).  N % 128 == 0, K % 256 == 0.  The kernel reads W in the PACKED layout
 * that samd_gemm_pack_weights writes once at load time (64 KiB blocks of 128 columns x 256 k in lane order, so that every
 * workgroup reads one linear stream).  splits == 1: writes d_out (dtype, [rows_pad][N]); splits > 1: writes fp32
 * partial sums d_partial [splits][rows_pad][N] that the consuming kernel adds up (samd_rmsnorm / samd_rope_kv_write /
 * samd_silu_mul with n_partials > 0). */
int samd_gemm_pack_weights(const void *d_W, void *d_packed, int32_t N, int32_t K, void *stream);
/* the MLP's gate and up projections with LlamaMLP's activation in the epilogue: d_out [rows_pad][N/2] =
 * silu(A @ Wgate^T) * (A @ Wup^T) with the model dtype's roundings (HF LlamaMLP.forward: act_fn(gate_proj(x)) * up_proj(x)).
 * d_W = samd_gemm_pack_weights of the [N][K] matrix whose rows are gate and up interleaved in groups of 64:
 * rows 128t..128t+63 = gate rows 64t..64t+63, rows 128t+64..128t+127 = up rows 64t..64t+63. */
int samd_gemm_skinny_silu(const void *d_A, const void *d_W, int32_t rows_pad, int32_t N, int32_t K, void *d_out, int32_t dtype,
                          void *stream);
int samd_gemm_splits(int32_t N, int32_t K, int32_t rows_pad);
int64_t samd_gemm_workspace(int32_t rows_pad, int32_t N, int32_t splits);
/* LlamaMLP's gate and up projections + act_fn(gate) * up on every CU: the unit of work is a pair of 16 gate columns and the 16 up
 * columns they multiply, dealt out evenly over one workgroup per CU (samd_gemm_skinny_silu's 128-column tiles leave a third of the
 * CUs idle for intermediate sizes like 11008).  d_Wg = the [gate; up] matrix with its rows interleaved in groups of 16 (gate rows
 * 16p.., then up rows 16p..), packed by samd_gemm_pack_groups (N = 2 * inter rows); out [rows_pad][inter].  Same arithmetic and
 * roundings as samd_gemm_skinny_silu.  Call site replaced: HF LlamaMLP.forward inside SO/samd_model.py:134-138. */
int samd_gemm_pack_groups(const void *d_W, void *d_packed, int32_t N, int32_t K, void *stream);
int samd_gemm_pairs_silu(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t inter, int32_t K, void *d_out, int32_t dtype, void *stream);
/* The q|k|v projection with RoPE and SamdStaticCache.update's row write (SO/cache.py:103-115) as its epilogue -- what samd_gemm_skinny
 * (q|k|v weights) + samd_rope_kv_write_cs do in two launches and a round trip of fp32 partial sums.  d_W64 = the [q|k|v] weight
 * matrix ((n_heads + 2 n_kv_heads) * 128 rows, K columns) packed by samd_gemm_pack_qkv64 (tiles of 24 or 32 complete rotate_half
 * pairs = 48 or 64 columns; the library picks the width from the matrix and the CU count, the same way in both calls); d_cs = samd_rope_rows' per-row cos | sin; q_out [rows][n_heads][128]; K / V rows go to
 * [n_kv_heads][max_len][128] caches at [L, L + n).  One workgroup per tile, no split-K: worth it when there are >= ~150 tiles
 * (a 32-head MHA model has 192; the runner keeps the two-launch path otherwise).  Call sites replaced: SO/samd_model.py:134-138
 * (the q/k/v projections, rotary embedding and cache update inside HF's LlamaAttention.forward). */
int samd_gemm_pack_qkv64(const void *d_W, void *d_packed, int32_t n_heads_total, int32_t K, void *stream);
/* The "norm-fold" forward at 16 rows (round 3): LlamaRMSNorm is applied by the projection that CONSUMES it, and the residual add by the
 * projection that PRODUCES the delta, so that a decoder layer is six launches instead of eight and no fp32 split-K partial is written:
 *   samd_embed_rows_ssq      rows of the embedding table -> x, plus d_ssq [hidden / 16][16] fp32: every row's sum of squares per 16-column
 *                            tile (tile-major);
 *   samd_gemm_qkv_rope_norm  samd_gemm_qkv_rope reading the RESIDUAL STREAM d_x [16][K]: 1 / rms per row from d_ssq (added up in a fixed
 *                            order by every workgroup), h = (x / rms).to(dtype), a = norm_weight * h on the way into LDS;
 *   samd_gemm_pairs_silu_norm  samd_gemm_pairs_silu in the same way (post_attention_layernorm + gate | up + SiLU * up);
 *                            both: rows_pad 16, or 8 (round 4): only rows 0..7 of d_x are fetched -- a draft of <= 8 nodes; rows 0..7 of the
 *                            results are bit-identical, rows 8..15 are those of a zero input;
 *   samd_gemm_cs_residual    x[m][n] <- (x[m][n] + dtype((A W^T)[m][n])).to(dtype) for o_proj / down_proj with COMPLETE sums (one workgroup per 16
 *                            output columns and all of K; d_Wg = the [N][K] matrix packed by samd_gemm_pack_groups), and the new
 *                            d_ssq [N / 16][16] of the updated rows.  rows_pad 16, or 8: only rows 0..7 of d_A are read and only rows 0..7 of
 *                            d_x / d_ssq written (a draft of <= 8 nodes: the launch pulls half the activation bytes).
 * Roundings are LlamaDecoderLayer's / LlamaRMSNorm's; 1 / rms may differ from samd_rmsnorm's in the last bit (another summation order).
 * Call sites replaced: SO/samd_model.py:134-138 (input_layernorm, post_attention_layernorm and the two residual adds of every
 * LlamaDecoderLayer inside the verify forward). */
/* samd_embed_rows_ssq + samd_rope_rows (head_dim 128) as ONE launch: the two open every norm-fold forward and do not depend on each
 * other (round 4; arguments as of the two) */
int samd_embed_rows_ssq_rope(const int32_t *d_tokens, const void *d_table, void *d_out, float *d_ssq, int32_t rows, int32_t hidden, int32_t vocab,
                             int32_t dtype, const int32_t *d_rel_pos, const int32_t *d_base, const float *d_cos, const float *d_sin, float *d_cs,
                             int32_t rope_rows, int32_t head_dim, int32_t max_pos, void *stream);
int samd_embed_rows_ssq(const int32_t *d_tokens, const void *d_table, void *d_out, float *d_ssq, int32_t rows, int32_t hidden, int32_t vocab,
                        int32_t dtype, void *stream);
int samd_gemm_qkv_rope_norm(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_W64, int32_t rows_pad, int32_t K,
                            const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_v_cache,
                            int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype, void *stream);
int samd_gemm_pairs_silu_norm(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_Wg, int32_t rows_pad, int32_t inter,
                              int32_t K, void *d_out, int32_t dtype, void *stream);
int samd_gemm_cs_residual(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t N, int32_t K, void *d_x, float *d_ssq, int32_t dtype, void *stream);
int samd_gemm_qkv_rope(const void *d_A, const void *d_W64, int32_t rows_pad, int32_t K, const float *d_cs, const int32_t *d_cache_length,
                       const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_v_cache, int32_t n_heads, int32_t n_kv_heads,
                       int32_t head_dim, int64_t max_len, int32_t dtype, void *stream);
/* round 6: the two above with the V rows written into a TRANSPOSED cache, d_vt_cache [H_kv][128][max_len] (samd_tree_attention_vt) */
int samd_gemm_qkv_rope_norm_vt(const void *d_x, const float *d_ssq, const void *d_norm_weight, float eps, const void *d_W64, int32_t rows_pad, int32_t K,
                               const float *d_cs, const int32_t *d_cache_length, const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_vt_cache,
                               int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int64_t max_len, int32_t dtype, void *stream);
int samd_gemm_qkv_rope_vt(const void *d_A, const void *d_W64, int32_t rows_pad, int32_t K, const float *d_cs, const int32_t *d_cache_length,
                          const int32_t *d_n, void *d_q_out, void *d_k_cache, void *d_vt_cache, int32_t n_heads, int32_t n_kv_heads,
                          int32_t head_dim, int64_t max_len, int32_t dtype, void *stream);
int samd_gemm_skinny(const void *d_A, const void *d_W, int32_t rows_pad, int32_t N, int32_t K, int32_t splits, float *d_partial,
                     void *d_out, int32_t dtype, void *stream);
/* round 6: the same product over a GROUP-MAJOR matrix (samd_gemm_pack_groups, the layout samd_gemm_cs_residual streams): o_proj / down_proj keep ONE
 * packed copy for every row bucket instead of a 128-column-tile copy for the split-K kernel next to the 16-column-group copy of the norm-fold
 * forward (-4 GB of a 7B replica).  Results are bit-identical to samd_gemm_skinny's. */
int samd_gemm_skinny_groups(const void *d_A, const void *d_Wg, int32_t rows_pad, int32_t N, int32_t K, int32_t splits, float *d_partial,
                            void *d_out, int32_t dtype, void *stream);

/* ---- scripted verifier (tests, smoke and bench only): replaces the LM arg-max of every draft node by
 * the next token of a target stream while the node's context (committed history + root->node path) is a
 * prefix of it, otherwise by a hash of the context's last three tokens -- the device twin of
 * tests/scripted_lm.py, so that acceptance statistics are reproducible without model weights.
 * d_target int32[n_target]; d_out int32[64]. */
int samd_scripted_argmax(samd_session_t *s, const int32_t *d_target, int32_t n_target, int32_t vocab, int32_t *d_out,
                         void *stream);

/* the same hook for callers that learn from the logits (bench.py --variant token_recycle): on top of the model's own row of draft node
 * i it writes the scripted arg-max as the best entry and, below it, the four successors of the node's two-token context in the bench's
 * sparse order-2 Markov source (bench._succ) in rank order -- a model that continues the text also ranks its plausible continuations.
 * d_argmax int32[64] (samd_scripted_argmax's output), d_logits [rows][row_stride] of dtype (rows of the draft beyond `rows` are
 * left alone).  Tests and bench only. */
int samd_scripted_logits(samd_session_t *s, const int32_t *d_argmax, void *d_logits, int32_t dtype, int32_t rows, int64_t row_stride,
                         int32_t markov_vocab, void *stream);
/* the same for a source whose next token depends on the LAST token alone (bench.py --variant token_recycle: bench._succ1, eight
 * successors per token over the ids [3, hot_vocab)): what a token-keyed successor table can learn, as the reference's does on natural
 * text (S/tree_model/token_recycle/token_recycle.py:40-48).  Tests and bench only. */
int samd_scripted_logits_order1(samd_session_t *s, const int32_t *d_argmax, void *d_logits, int32_t dtype, int32_t rows, int64_t row_stride,
                                int32_t hot_vocab, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Token Recycle (S/tree_model/token_recycle/token_recycle.py:18-63)
 * ---------------------------------------------------------------------------------------------- */
int samd_recycle_create(int32_t vocab, const int32_t *h_child_offsets, const int32_t *h_children, int32_t n_nodes,
                        samd_recycle_t **out);
void samd_recycle_free(samd_recycle_t *t);
/* update: logits.topk(8) of every verified token, cache[token] = top-8, later rows win (:40-48) */
int samd_recycle_update(samd_recycle_t *t, const int32_t *d_tokens, const void *d_logits, int32_t dtype, int32_t n,
                        const int32_t *d_n, int64_t vocab, int64_t row_stride, void *stream);
/* gen_draft: fill the static tree from the table (:50-60); d_out int32[n_nodes] */
int samd_recycle_draft(samd_recycle_t *t, const int32_t *d_start_token, int32_t *d_out, void *stream);
int samd_recycle_export(samd_recycle_t *t, int32_t *h_table, uint8_t *h_present, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMD_HIP_H */
