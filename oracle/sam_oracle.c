/*
 * sam_oracle.c -- CPU restatement of SAM-Decoding's draft+verify integer path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
 * sam-decoding_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product never links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked against fixtures under
 * tests/golden/ that were produced by importing the Python reference itself in the
 * dev container (tests/golden/make_golden.py); see tests/test_oracle_golden.py.
 * (The reference ships no tests/golden vectors of its own -- SURVEY.md section 4.)
 *
 * Citations are relative to /root/reference.  SO/ = samd_sam_only/, S/ = samd/.
 *
 * Plain C99, single thread (the reference is single-threaded Python).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* growable arrays + (state,token) -> edge-id hash                            */
/* ------------------------------------------------------------------------- */

typedef struct {
    /* per state (SO/sam/static_sam.py:24-29, SO/sam/dyn_sam.py:13-18) */
    int32_t *link, *length, *aux;   /* aux = cnt_endpos (kind 0) or min_endpos (kind 1) */
    int32_t *head, *tail, *deg;     /* insertion-ordered edge list == dict order of `next` */
    int64_t n_states, cap_states;
    /* edge pool */
    int32_t *e_tok, *e_dst, *e_next;
    int64_t n_edges, cap_edges;
    /* hash: key (state,tok) -> edge id */
    uint64_t *h_key; int32_t *h_val; uint64_t h_mask; int64_t h_used;
    /* text: input_ids with -1 sentinel at [0]  (SO/sam/dyn_sam.py:27-28) */
    int32_t *text; int64_t n_text, cap_text;
    /* top-k next table [n_states][8] (tok,dst), -1 padded  (SO/sam/static_sam.py:137-146) */
    int32_t *topk_tok, *topk_dst, *topk_n; int64_t topk_states;
    int32_t kind;                   /* 0: SO static (counts, no text). 1: dyn / S-static (min_endpos + text) */
    int32_t last, max_length, cur_index, cur_length;
} osam_t;

#define HEMPTY 0xFFFFFFFFFFFFFFFFull

static uint64_t hmix(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

static void h_alloc(osam_t *s, uint64_t size) {
    s->h_key = (uint64_t *)malloc(size * sizeof(uint64_t));
    s->h_val = (int32_t *)malloc(size * sizeof(int32_t));
    memset(s->h_key, 0xFF, size * sizeof(uint64_t));
    s->h_mask = size - 1; s->h_used = 0;
}

static void h_put_raw(osam_t *s, uint64_t key, int32_t val) {
    uint64_t i = hmix(key) & s->h_mask;
    while (s->h_key[i] != HEMPTY) i = (i + 1) & s->h_mask;
    s->h_key[i] = key; s->h_val[i] = val; s->h_used++;
}

static void h_grow(osam_t *s) {
    uint64_t *ok = s->h_key; int32_t *ov = s->h_val; uint64_t osz = s->h_mask + 1;
    h_alloc(s, osz * 2);
    for (uint64_t i = 0; i < osz; i++) if (ok[i] != HEMPTY) h_put_raw(s, ok[i], ov[i]);
    free(ok); free(ov);
}

static inline uint64_t ekey(int32_t state, int32_t tok) { return ((uint64_t)(uint32_t)state << 32) | (uint32_t)tok; }

/* edge id of `tok in states[state].next`, or -1 */
static int32_t edge_find(const osam_t *s, int32_t state, int32_t tok) {
    uint64_t key = ekey(state, tok), i = hmix(key) & s->h_mask;
    while (s->h_key[i] != HEMPTY) {
        if (s->h_key[i] == key) return s->h_val[i];
        i = (i + 1) & s->h_mask;
    }
    return -1;
}

/* states[state].next[tok] = dst for a NEW key: appended at the end of the dict order */
static void edge_insert(osam_t *s, int32_t state, int32_t tok, int32_t dst) {
    if (s->n_edges == s->cap_edges) {
        s->cap_edges *= 2;
        s->e_tok = (int32_t *)realloc(s->e_tok, s->cap_edges * sizeof(int32_t));
        s->e_dst = (int32_t *)realloc(s->e_dst, s->cap_edges * sizeof(int32_t));
        s->e_next = (int32_t *)realloc(s->e_next, s->cap_edges * sizeof(int32_t));
    }
    int32_t e = (int32_t)s->n_edges++;
    s->e_tok[e] = tok; s->e_dst[e] = dst; s->e_next[e] = -1;
    if (s->head[state] < 0) s->head[state] = e; else s->e_next[s->tail[state]] = e;
    s->tail[state] = e; s->deg[state]++;
    if ((uint64_t)(s->h_used + 1) * 2 > s->h_mask + 1) h_grow(s);
    h_put_raw(s, ekey(state, tok), e);
}

static int32_t state_new(osam_t *s, int32_t link, int32_t length, int32_t aux) {
    if (s->n_states == s->cap_states) {
        s->cap_states *= 2;
        size_t b = s->cap_states * sizeof(int32_t);
        s->link = (int32_t *)realloc(s->link, b); s->length = (int32_t *)realloc(s->length, b);
        s->aux = (int32_t *)realloc(s->aux, b); s->head = (int32_t *)realloc(s->head, b);
        s->tail = (int32_t *)realloc(s->tail, b); s->deg = (int32_t *)realloc(s->deg, b);
    }
    int32_t i = (int32_t)s->n_states++;
    s->link[i] = link; s->length[i] = length; s->aux[i] = aux;
    s->head[i] = s->tail[i] = -1; s->deg[i] = 0;
    return i;
}

static void sam_init_root(osam_t *s) {
    /* root: next={}, link=-1, length=0, aux=0   (SO/sam/static_sam.py:48, SO/sam/dyn_sam.py:26) */
    s->n_states = 0; s->n_edges = 0;
    memset(s->h_key, 0xFF, (s->h_mask + 1) * sizeof(uint64_t)); s->h_used = 0;
    state_new(s, -1, 0, 0);
    s->n_text = 0; s->text[s->n_text++] = -1;     /* input_ids = [-1] */
    s->last = 0; s->max_length = 0; s->cur_index = 0; s->cur_length = 0;
}

osam_t *osam_new(int32_t kind) {
    osam_t *s = (osam_t *)calloc(1, sizeof(osam_t));
    s->kind = kind;
    s->cap_states = 1024; s->cap_edges = 1024; s->cap_text = 1024;
    size_t b = s->cap_states * sizeof(int32_t);
    s->link = (int32_t *)malloc(b); s->length = (int32_t *)malloc(b); s->aux = (int32_t *)malloc(b);
    s->head = (int32_t *)malloc(b); s->tail = (int32_t *)malloc(b); s->deg = (int32_t *)malloc(b);
    s->e_tok = (int32_t *)malloc(s->cap_edges * 4); s->e_dst = (int32_t *)malloc(s->cap_edges * 4);
    s->e_next = (int32_t *)malloc(s->cap_edges * 4);
    s->text = (int32_t *)malloc(s->cap_text * 4);
    h_alloc(s, 4096);
    sam_init_root(s);
    return s;
}

void osam_free(osam_t *s) {
    if (!s) return;
    free(s->link); free(s->length); free(s->aux); free(s->head); free(s->tail); free(s->deg);
    free(s->e_tok); free(s->e_dst); free(s->e_next); free(s->h_key); free(s->h_val); free(s->text);
    free(s->topk_tok); free(s->topk_dst); free(s->topk_n); free(s);
}

/* DynSAM.reset: discard the automaton  (SO/sam/dyn_sam.py:37-43) */
void osam_reset_all(osam_t *s) { sam_init_root(s); }
/* StaticSAM.reset: rewind the cursor only  (SO/sam/static_sam.py:127-129) */
void osam_reset_cursor(osam_t *s) { s->cur_index = 0; s->cur_length = 0; }

/* ------------------------------------------------------------------------- */
/* transfer_state  (SO/sam/static_sam.py:98-107 == SO/sam/dyn_sam.py:78-87    */
/*                  == S/sam/static_sam.py:81-90 == S/sam/dyn_sam.py:69-78)   */
/* ------------------------------------------------------------------------- */
void osam_transfer_state(const osam_t *s, int32_t index, int32_t length, int32_t tok,
                         int32_t *o_index, int32_t *o_length) {
    int32_t e;
    while (index != 0 && (e = edge_find(s, index, tok)) < 0) {
        index = s->link[index];
        length = s->length[index];
    }
    e = edge_find(s, index, tok);
    if (e >= 0) { index = s->e_dst[e]; length += 1; }
    else { index = 0; length = 0; }
    *o_index = index; *o_length = length;
}

/* lookup: peek, no commit  (SO/sam/static_sam.py:122-125, SO/sam/dyn_sam.py:111-114) */
void osam_lookup(const osam_t *s, int32_t tok, int32_t *o_index, int32_t *o_length) {
    osam_transfer_state(s, s->cur_index, s->cur_length, tok, o_index, o_length);
}

/* transfer_tokens: commit the cursor  (SO/sam/static_sam.py:118-120) */
void osam_transfer_tokens(osam_t *s, const int32_t *toks, int64_t n) {
    for (int64_t i = 0; i < n; i++)
        osam_transfer_state(s, s->cur_index, s->cur_length, toks[i], &s->cur_index, &s->cur_length);
}

/* ------------------------------------------------------------------------- */
/* add_state  (SO/sam/static_sam.py:67-96 [counts], SO/sam/dyn_sam.py:50-76,  */
/*             S/sam/static_sam.py:53-79, S/sam/dyn_sam.py:41-67 [min_endpos])*/
/* ------------------------------------------------------------------------- */
static void add_state(osam_t *s, int32_t tok) {
    s->max_length += 1;
    int32_t cur = state_new(s, -1, s->max_length, s->kind == 0 ? 0 : s->max_length);
    int32_t p = s->last, e = -1;
    while (p != -1 && (e = edge_find(s, p, tok)) < 0) {
        edge_insert(s, p, tok, cur);
        p = s->link[p];
    }
    if (p == -1) {
        s->link[cur] = 0;
    } else {
        int32_t q = s->e_dst[e];
        if (s->length[p] + 1 == s->length[q]) {
            s->link[cur] = q;
        } else {
            /* clone = deepcopy(q): same dict (order kept), link, aux; length = len(p)+1 */
            int32_t clone = state_new(s, s->link[q], s->length[p] + 1, s->aux[q]);
            for (int32_t qe = s->head[q]; qe >= 0; qe = s->e_next[qe])
                edge_insert(s, clone, s->e_tok[qe], s->e_dst[qe]);
            while (p != -1) {
                int32_t pe = edge_find(s, p, tok);      /* key exists along this chain */
                if (pe < 0 || s->e_dst[pe] != q) break;
                s->e_dst[pe] = clone;                  /* re-point: dict order unchanged */
                p = s->link[p];
            }
            s->link[q] = clone; s->link[cur] = clone;
        }
    }
    s->last = cur;
    if (s->kind == 0) {                                /* SO/sam/static_sam.py:94-96 */
        while (cur != 0) { s->aux[cur] += 1; cur = s->link[cur]; }
    }
}

/* add_tokens: transfer the cursor FIRST, then extend  (SO/sam/dyn_sam.py:101-105,
 * SO/sam/static_sam.py:113-116); text kinds append to input_ids afterwards. */
void osam_add_tokens(osam_t *s, const int32_t *toks, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        osam_transfer_state(s, s->cur_index, s->cur_length, toks[i], &s->cur_index, &s->cur_length);
        add_state(s, toks[i]);
    }
    if (s->kind == 1) {
        while (s->n_text + n > s->cap_text) {
            s->cap_text *= 2; s->text = (int32_t *)realloc(s->text, s->cap_text * 4);
        }
        memcpy(s->text + s->n_text, toks, n * 4); s->n_text += n;
    }
}

/* add_batch_tokens: one automaton for all docs, EOS appended when missing
 * (SO/sam/static_sam.py:131-135, S/sam/static_sam.py:31-35) */
void osam_add_batch(osam_t *s, const int32_t *toks, const int64_t *doc_off, int64_t n_docs, int32_t eos) {
    for (int64_t d = 0; d < n_docs; d++) {
        int64_t b = doc_off[d], e = doc_off[d + 1];
        osam_add_tokens(s, toks + b, e - b);
        if (toks[e - 1] != eos) osam_add_tokens(s, &eos, 1);
    }
}

/* init_topk_next: stable descending sort of dict items by child count, first k<=8
 * (SO/sam/static_sam.py:137-146) */
void osam_init_topk(osam_t *s) {
    const int K = 8;
    free(s->topk_tok); free(s->topk_dst); free(s->topk_n);
    s->topk_states = s->n_states;
    s->topk_tok = (int32_t *)malloc(s->n_states * K * 4);
    s->topk_dst = (int32_t *)malloc(s->n_states * K * 4);
    s->topk_n = (int32_t *)malloc(s->n_states * 4);
    for (int64_t st = 0; st < s->n_states; st++) {
        int32_t *tt = s->topk_tok + st * K, *td = s->topk_dst + st * K; int m = 0;
        /* insertion into a sorted prefix keeps equal keys in dict order (stable, reverse=True) */
        for (int32_t e = s->head[st]; e >= 0; e = s->e_next[e]) {
            int32_t c = s->aux[s->e_dst[e]];
            int pos = m;
            while (pos > 0 && s->aux[td[pos - 1]] < c) pos--;
            if (pos >= K) continue;
            int last = m < K ? m : K - 1;
            for (int j = last; j > pos; j--) { tt[j] = tt[j - 1]; td[j] = td[j - 1]; }
            tt[pos] = s->e_tok[e]; td[pos] = s->e_dst[e];
            if (m < K) m++;
        }
        for (int j = m; j < K; j++) { tt[j] = -1; td[j] = -1; }
        s->topk_n[st] = m;
    }
}

/* ------------------------------------------------------------------------- */
/* accessors used by the tests and by the golden comparison                   */
/* ------------------------------------------------------------------------- */
int64_t osam_num_states(const osam_t *s) { return s->n_states; }
int64_t osam_num_edges(const osam_t *s) { return s->n_edges; }
int64_t osam_text_len(const osam_t *s) { return s->n_text; }
void osam_cursor(const osam_t *s, int32_t *idx, int32_t *len) { *idx = s->cur_index; *len = s->cur_length; }
void osam_set_cursor(osam_t *s, int32_t idx, int32_t len) { s->cur_index = idx; s->cur_length = len; }
int32_t osam_last(const osam_t *s) { return s->last; }
int32_t osam_max_length(const osam_t *s) { return s->max_length; }

void osam_export_states(const osam_t *s, int32_t *link, int32_t *length, int32_t *aux, int32_t *deg) {
    memcpy(link, s->link, s->n_states * 4); memcpy(length, s->length, s->n_states * 4);
    memcpy(aux, s->aux, s->n_states * 4); memcpy(deg, s->deg, s->n_states * 4);
}
/* edges of all states, state-major, each state's edges in dict (insertion) order */
void osam_export_edges(const osam_t *s, int32_t *tok, int32_t *dst) {
    int64_t k = 0;
    for (int64_t st = 0; st < s->n_states; st++)
        for (int32_t e = s->head[st]; e >= 0; e = s->e_next[e]) { tok[k] = s->e_tok[e]; dst[k] = s->e_dst[e]; k++; }
}
void osam_export_text(const osam_t *s, int32_t *text) { memcpy(text, s->text, s->n_text * 4); }
void osam_export_topk(const osam_t *s, int32_t *tok, int32_t *dst, int32_t *n) {
    memcpy(tok, s->topk_tok, s->topk_states * 8 * 4); memcpy(dst, s->topk_dst, s->topk_states * 8 * 4);
    memcpy(n, s->topk_n, s->topk_states * 4);
}

/* ------------------------------------------------------------------------- */
/* drafts                                                                     */
/* ------------------------------------------------------------------------- */
static int32_t draft_size(int32_t match, double alpha, int32_t max_predicts) {
    /* n = min(max_predicts, 1 + int(match_length * alpha))   (SO/sam/dyn_sam.py:117) */
    int32_t n = 1 + (int32_t)((double)match * alpha);
    return n < max_predicts ? n : max_predicts;
}

/* DynSAM.gen_draft (sam_only): [start] + input_ids[e+1 : e+n]  (SO/sam/dyn_sam.py:116-121) */
int32_t osam_gen_draft_seq(const osam_t *s, int32_t index, int32_t match, int32_t start,
                           int32_t max_predicts, double alpha, int32_t *out) {
    int32_t n = draft_size(match, alpha, max_predicts);
    int64_t e = s->aux[index], lo = e + 1, hi = e + n;
    if (hi > s->n_text) hi = s->n_text;
    int32_t m = 0; out[m++] = start;
    for (int64_t i = lo; i < hi; i++) out[m++] = s->text[i];
    return m;
}

/* full-variant DynSAM.to_anc  (S/sam/dyn_sam.py:99-105) */
int32_t osam_to_anc(const osam_t *s, int32_t index, int32_t n_predicts) {
    if (index != 0) {
        int32_t to_end = s->max_length - s->aux[index];
        while (s->link[index] != 0 && n_predicts > to_end) {
            index = s->link[index];
            to_end = s->max_length - s->aux[index];
        }
    }
    return index;
}

/* full-variant gen_draft: fixed n_predicts tokens, zero padded
 * (S/sam/dyn_sam.py:107-113 with to_anc; S/sam/static_sam.py:119-125 without) */
void osam_gen_draft_fixed(const osam_t *s, int32_t index, int32_t start, int32_t n_predicts,
                          int32_t use_to_anc, int32_t *out) {
    if (use_to_anc) index = osam_to_anc(s, index, n_predicts);
    int64_t e = s->aux[index], lo = e + 1, hi = e + n_predicts;
    if (hi > s->n_text) hi = s->n_text;
    int32_t m = 0; out[m++] = start;
    for (int64_t i = lo; i < hi && m < n_predicts; i++) out[m++] = s->text[i];
    while (m < n_predicts) out[m++] = 0;
}

/* StaticSAM.gen_draft (sam_only tree)  (SO/sam/static_sam.py:13-19, :182-215).
 * CPython heapq order is reproduced exactly: heappush = append + _siftdown,
 * heappop = move last to root, _siftup to a leaf taking the right child when
 * `not left < right`, then _siftdown.  Ordering key = prob (IEEE double) only. */
typedef struct { double prob; int32_t token, index, anc, depth; } oitem_t;

static void hq_siftdown(oitem_t *h, int start, int pos) {
    oitem_t x = h[pos];
    while (pos > start) {
        int par = (pos - 1) >> 1;
        if (x.prob < h[par].prob) { h[pos] = h[par]; pos = par; continue; }
        break;
    }
    h[pos] = x;
}
static void hq_push(oitem_t *h, int *n, oitem_t x) { h[*n] = x; (*n)++; hq_siftdown(h, 0, *n - 1); }
static oitem_t hq_pop(oitem_t *h, int *n) {
    oitem_t last = h[--(*n)];
    if (*n == 0) return last;
    oitem_t ret = h[0]; h[0] = last;
    int end = *n, pos = 0, child = 1; oitem_t x = h[0];
    while (child < end) {
        int right = child + 1;
        if (right < end && !(h[child].prob < h[right].prob)) child = right;
        h[pos] = h[child]; pos = child; child = 2 * pos + 1;
    }
    h[pos] = x;
    hq_siftdown(h, 0, pos);
    return ret;
}

int32_t osam_gen_draft_tree(const osam_t *s, int32_t index, int32_t match, int32_t start,
                            int32_t max_predicts, double alpha, int32_t K,
                            int32_t *tree, int32_t *anc_tree) {
    int32_t n = draft_size(match, alpha, max_predicts);
    if (n < 0) n = 0;
    oitem_t *h = (oitem_t *)malloc(sizeof(oitem_t) * (size_t)(8 * (n + 1) + 2));
    int32_t *dep_cnt = (int32_t *)calloc((size_t)n + 2, 4);
    int hn = 0, m = 0;
    oitem_t root = { -1.0, start, index, -1, 0 };
    hq_push(h, &hn, root);
    while (m != n && hn != 0) {
        oitem_t it = hq_pop(h, &hn);
        if (dep_cnt[it.depth] + 1 > K) continue;
        dep_cnt[it.depth] += 1;
        int32_t cur = m;
        tree[m] = it.token; anc_tree[m] = it.anc; m++;
        if (m == n) break;
        int32_t cnt_sum = s->aux[it.index];
        int32_t kk = s->topk_n[it.index]; if (kk > K) kk = K;   /* states_topk_next[i][:K] */
        for (int j = 0; j < kk; j++) {
            int32_t ni = s->topk_dst[(int64_t)it.index * 8 + j];
            double n_prob = (double)s->aux[ni] / (double)cnt_sum;   /* Python int / int */
            oitem_t c = { it.prob * n_prob, s->topk_tok[(int64_t)it.index * 8 + j], ni, cur, it.depth + 1 };
            hq_push(h, &hn, c);
        }
    }
    free(h); free(dep_cnt);
    return m;
}

/* gen_buffers(anc_tree)  (SO/sam/static_sam.py:148-180).
 * pos[n] = depth; mask[n*n] row i true on i and its ancestors; retrieve[leaves][max_depth]
 * root->leaf per leaf in increasing node order, -1 padded.  retrieve must hold n*n entries. */
void o_gen_buffers(const int32_t *anc, int32_t n, int64_t *pos, uint8_t *mask, int64_t *retrieve,
                   int32_t *n_leaves, int32_t *max_depth) {
    uint8_t *is_leaf = (uint8_t *)malloc((size_t)n);
    memset(is_leaf, 1, (size_t)n); memset(mask, 0, (size_t)n * n);
    pos[0] = 0;
    for (int i = 1; i < n; i++) { is_leaf[anc[i]] = 0; pos[i] = pos[anc[i]] + 1; }
    for (int i = 0; i < n; i++) for (int j = i; j != -1; j = anc[j]) mask[(size_t)i * n + j] = 1;
    int md = 0, nl = 0;
    for (int i = 0; i < n; i++) if (is_leaf[i] && pos[i] + 1 > md) md = (int)pos[i] + 1;
    for (int i = 0; i < n; i++) {
        if (!is_leaf[i]) continue;
        int64_t *row = retrieve + (size_t)nl * md;
        for (int j = 0; j < md; j++) row[j] = -1;
        for (int j = i; j != -1; j = anc[j]) row[pos[j]] = j;
        nl++;
    }
    *n_leaves = nl; *max_depth = md;
    free(is_leaf);
}

/* Token-Recycle static-tree buffers  (S/tree_model/token_recycle/utils.py:37-99):
 * child lists -> parent array, depth, mask, leaf rows in REVERSED node order. */
void o_tr_gen_buffers(const int32_t *child_off, const int32_t *childs, int32_t n, int32_t *anc,
                      int64_t *pos, uint8_t *mask, int64_t *retrieve, int32_t *n_leaves, int32_t *max_depth) {
    anc[0] = -1;
    for (int i = 0; i < n; i++) for (int c = child_off[i]; c < child_off[i + 1]; c++) anc[childs[c]] = i;
    pos[0] = 0;
    for (int i = 1; i < n; i++) pos[i] = pos[anc[i]] + 1;
    memset(mask, 0, (size_t)n * n);
    for (int i = 0; i < n; i++) for (int j = i; j != -1; j = anc[j]) mask[(size_t)i * n + j] = 1;
    int md = 0; for (int i = 0; i < n; i++) if (pos[i] + 1 > md) md = (int)pos[i] + 1;   /* max level + 1 */
    int nl = 0;
    for (int i = n - 1; i >= 0; i--) {
        if (child_off[i + 1] != child_off[i]) continue;
        int64_t *row = retrieve + (size_t)nl * md;
        for (int j = 0; j < md; j++) row[j] = -1;
        for (int j = i; j != -1; j = anc[j]) row[pos[j]] = j;
        nl++;
    }
    *n_leaves = nl; *max_depth = md;
}

/* ------------------------------------------------------------------------- */
/* verify side: candidates, greedy posterior                                  */
/* ------------------------------------------------------------------------- */
/* first-max argmax of a row (torch.argmax tie rule: lowest index) */
static int32_t argmax_f32(const float *x, int64_t v) {
    int64_t b = 0; float m = x[0];
    for (int64_t i = 1; i < v; i++) if (x[i] > m) { m = x[i]; b = i; }
    return (int32_t)b;
}
void o_argmax_rows(const float *logits, int64_t rows, int64_t v, int32_t *out) {
    for (int64_t r = 0; r < rows; r++) out[r] = argmax_f32(logits + r * v, v);
}

/* gen_candidates tail: candidate_tokens = (tokens + [0])[retrieve]  (SO/utils.py:94-97);
 * a -1 index selects the appended pad token 0. */
void o_candidates(const int32_t *tokens, int32_t n, const int64_t *retrieve, int32_t rows, int32_t depth,
                  int32_t *cand) {
    for (int i = 0; i < rows * depth; i++) { int64_t r = retrieve[i]; cand[i] = r < 0 ? 0 : tokens[r]; }
    (void)n;
}

/* eval_posterior, greedy branch  (SO/utils.py:127-141), on per-node arg-max tokens.
 * Equivalent to the reference's gathered form: logits[0][retrieve] with index -1 selects the
 * LAST tree node  (SO/samd_model.py:144).  For a sequence draft pass retrieve = NULL.
 * Outputs: best row, accept_length(+1, root included), and the node whose logits are the
 * next sample_p (logits[best, accept]). */
void o_eval_posterior(const int32_t *node_argmax, const int32_t *tokens, int32_t n,
                      const int64_t *retrieve, int32_t rows, int32_t depth,
                      int32_t *o_best, int32_t *o_accept, int32_t *o_next_node) {
    int32_t best = 0, best_acc = 0;
    if (!retrieve) { rows = 1; depth = n; }
    for (int r = 0; r < rows; r++) {
        int acc = 0;
        for (int j = 1; j < depth; j++) {
            int64_t cj = retrieve ? retrieve[(size_t)r * depth + j] : j;
            int64_t pj = retrieve ? retrieve[(size_t)r * depth + j - 1] : j - 1;
            int32_t cand = cj < 0 ? 0 : tokens[cj];
            int32_t am = node_argmax[pj < 0 ? n - 1 : pj];
            if (cand != am) break;
            acc++;
        }
        if (acc > best_acc) { best_acc = acc; best = r; }     /* first max */
    }
    int64_t nn = retrieve ? retrieve[(size_t)best * depth + best_acc] : best_acc;
    *o_best = best; *o_accept = best_acc + 1; *o_next_node = (int32_t)(nn < 0 ? n - 1 : nn);
}

/* ------------------------------------------------------------------------- */
/* Token Recycle table  (S/tree_model/token_recycle/token_recycle.py:33-60)   */
/* table[V][8], present[V].  update: later entries overwrite earlier ones.    */
/* ------------------------------------------------------------------------- */
void o_tr_update(int32_t *table, uint8_t *present, const int32_t *tree_tokens, const int32_t *topk, int32_t n) {
    for (int i = 0; i < n; i++) {
        memcpy(table + (size_t)tree_tokens[i] * 8, topk + (size_t)i * 8, 32);
        present[tree_tokens[i]] = 1;
    }
}
void o_tr_gen_draft(const int32_t *table, const uint8_t *present, const int32_t *child_off,
                    const int32_t *childs, int32_t n, int32_t start, int32_t *out) {
    memset(out, 0, (size_t)n * 4); out[0] = start;
    for (int i = 0; i < n; i++) {
        int32_t t = out[i];
        if (!present[t]) continue;
        for (int c = child_off[i]; c < child_off[i + 1]; c++) out[childs[c]] = table[(size_t)t * 8 + (c - child_off[i])];
    }
}
/* logits.topk(8).indices: descending values; ties -> lower index first is what torch CPU
 * returns for the fixtures used here (tests only use tie-free rows). */
void o_topk8_rows(const float *logits, int64_t rows, int64_t v, int32_t *out) {
    for (int64_t r = 0; r < rows; r++) {
        const float *x = logits + r * v; int32_t idx[8]; int m = 0;
        for (int64_t i = 0; i < v; i++) {
            int pos = m;
            while (pos > 0 && x[idx[pos - 1]] < x[i]) pos--;
            if (pos >= 8) continue;
            int last = m < 8 ? m : 7;
            for (int j = last; j > pos; j--) idx[j] = idx[j - 1];
            idx[pos] = (int32_t)i; if (m < 8) m++;
        }
        memcpy(out + r * 8, idx, 32);
    }
}

/* ------------------------------------------------------------------------- */
/* one sam_only draft step on the CPU: DraftModel.lookup + gen_buffers        */
/* (SO/draft.py:50-59).  Returns 0 = sequence, 1 = tree.  Used by tests and   */
/* as the timed cpu_baseline "port" in bench.py.                              */
/* ------------------------------------------------------------------------- */
int32_t o_draft_lookup_so(const osam_t *dyn, const osam_t *stat, int32_t start, int32_t max_predicts,
                          double alpha, int32_t K, int32_t len_bias, int32_t *tokens, int32_t *anc, int32_t *n_out) {
    int32_t id, md, is, ms;
    osam_lookup(dyn, start, &id, &md);
    osam_lookup(stat, start, &is, &ms);
    ms -= len_bias;
    if (md >= ms) {
        int32_t m = osam_gen_draft_seq(dyn, id, md, start, max_predicts, alpha, tokens);
        for (int i = 0; i < m; i++) anc[i] = i - 1;
        *n_out = m; return 0;
    }
    *n_out = osam_gen_draft_tree(stat, is, ms, start, max_predicts, alpha, K, tokens, anc);
    return 1;
}

/* DraftModel.update  (SO/draft.py:62-67) */
void o_draft_update(osam_t *dyn, osam_t *stat, const int32_t *toks, int32_t n) {
    osam_add_tokens(dyn, toks, n);
    osam_transfer_tokens(stat, toks, n);
}
