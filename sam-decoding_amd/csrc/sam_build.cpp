// sam_build.cpp -- host-side construction of the static (corpus) suffix automaton and its HBM image.
//
// Replaces StaticSAM.build / add_batch_tokens / add_state / init_topk_next
// (reference: samd_sam_only/sam/static_sam.py:31-40, :67-96, :131-146; samd/sam/static_sam.py:31-79)
// and dump_sam / load_sam (samd_sam_only/sam/utils.py:20-39).  Construction is inherently sequential
// and offline in the reference (tools/gen_sam_alpaca_sam_only.py), so it stays on the host; what it
// emits is the flat 64-byte-node image that the gfx950 kernels walk (samd_common.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>
#include "samd_common.h"

static thread_local char g_err[512] = "";
static thread_local char g_detail[256] = "";
// a detail recorded below the place that names the failing entry point (e.g. what the device's LDS offers): the next samd_set_error
// appends it instead of overwriting it
void samd_set_error_detail(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_detail, sizeof(g_detail), fmt, ap); va_end(ap);
}
void samd_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    if (g_detail[0]) {
        const size_t n = strlen(g_err);
        snprintf(g_err + n, sizeof(g_err) - n, " [%s]", g_detail);
        g_detail[0] = 0;
    }
}
extern "C" const char *samd_last_error(void) { return g_err; }

namespace {

// Online suffix-automaton builder.  Transitions live in one edge pool; every state threads its edges
// in first-insertion order (the reference's dict order, which decides top-k ties) and a single
// open-addressing index maps (state, token) -> edge.
class Builder {
public:
    explicit Builder(int kind) : kind_(kind) {
        index_key_.assign(1u << 16, ~0ull); index_val_.assign(1u << 16, -1); index_mask_ = (1u << 16) - 1;
        new_state(-1, 0, 0);
        text_.push_back(-1);
    }

    void add_document(const int32_t *tok, int64_t n, int32_t eos) {
        for (int64_t i = 0; i < n; i++) extend(tok[i]);
        if (kind_ == SAMD_KIND_ENDPOS) text_.insert(text_.end(), tok, tok + n);
        if (n > 0 && tok[n - 1] != eos) { extend(eos); if (kind_ == SAMD_KIND_ENDPOS) text_.push_back(eos); }
    }

    int kind_;
    std::vector<int32_t> link_, length_, aux_, first_, lastedge_, deg_;
    std::vector<int32_t> edge_tok_, edge_dst_, edge_sibling_;
    std::vector<int32_t> text_;

private:
    std::vector<uint64_t> index_key_;
    std::vector<int32_t> index_val_;
    uint64_t index_mask_;
    size_t index_used_ = 0;
    int32_t tip_ = 0, text_len_ = 0;

    static uint64_t scramble(uint64_t k) {
        k *= 0x9E3779B97F4A7C15ull; k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32;
        return k;
    }
    static uint64_t pack(int32_t s, int32_t t) { return ((uint64_t)(uint32_t)s << 32) | (uint32_t)t; }

    int32_t new_state(int32_t link, int32_t length, int32_t aux) {
        link_.push_back(link); length_.push_back(length); aux_.push_back(aux);
        first_.push_back(-1); lastedge_.push_back(-1); deg_.push_back(0);
        return (int32_t)link_.size() - 1;
    }
    int32_t find_edge(int32_t s, int32_t t) const {
        uint64_t key = pack(s, t);
        for (uint64_t i = scramble(key) & index_mask_;; i = (i + 1) & index_mask_) {
            if (index_key_[i] == key) return index_val_[i];
            if (index_key_[i] == ~0ull) return -1;
        }
    }
    void index_put(uint64_t key, int32_t val) {
        uint64_t i = scramble(key) & index_mask_;
        while (index_key_[i] != ~0ull) i = (i + 1) & index_mask_;
        index_key_[i] = key; index_val_[i] = val; index_used_++;
    }
    void grow_index() {
        std::vector<uint64_t> ok; std::vector<int32_t> ov;
        ok.swap(index_key_); ov.swap(index_val_);
        size_t nsz = ok.size() * 2;
        index_key_.assign(nsz, ~0ull); index_val_.assign(nsz, -1); index_mask_ = nsz - 1; index_used_ = 0;
        for (size_t i = 0; i < ok.size(); i++) if (ok[i] != ~0ull) index_put(ok[i], ov[i]);
    }
    void append_edge(int32_t s, int32_t t, int32_t dst) {
        int32_t e = (int32_t)edge_tok_.size();
        edge_tok_.push_back(t); edge_dst_.push_back(dst); edge_sibling_.push_back(-1);
        if (first_[s] < 0) first_[s] = e; else edge_sibling_[lastedge_[s]] = e;
        lastedge_[s] = e; deg_[s]++;
        if ((index_used_ + 1) * 2 > index_key_.size()) grow_index();
        index_put(pack(s, t), e);
    }

    void extend(int32_t t) {
        text_len_++;
        int32_t cur = new_state(-1, text_len_, kind_ == SAMD_KIND_COUNT ? 0 : text_len_);
        int32_t p = tip_, e = -1;
        for (; p != -1; p = link_[p]) {
            e = find_edge(p, t);
            if (e >= 0) break;
            append_edge(p, t, cur);
        }
        if (p == -1) {
            link_[cur] = 0;
        } else {
            int32_t q = edge_dst_[e];
            if (length_[p] + 1 == length_[q]) {
                link_[cur] = q;
            } else {
                // split q: the copy keeps q's transitions (same order), suffix link and aux value
                int32_t cp = new_state(link_[q], length_[p] + 1, aux_[q]);
                for (int32_t qe = first_[q]; qe >= 0; qe = edge_sibling_[qe]) append_edge(cp, edge_tok_[qe], edge_dst_[qe]);
                for (; p != -1; p = link_[p]) {
                    int32_t pe = find_edge(p, t);
                    if (pe < 0 || edge_dst_[pe] != q) break;
                    edge_dst_[pe] = cp;
                }
                link_[q] = cp; link_[cur] = cp;
            }
        }
        tip_ = cur;
        if (kind_ == SAMD_KIND_COUNT)
            for (int32_t v = cur; v != 0; v = link_[v]) aux_[v]++;    // occurrence counts along the suffix chain
    }
};

// SAMD_RUN flags (samd_common.h): run[s] = number of consecutive states s, s + 1, ... whose rank-0 successor is the next state
// in memory (capped at 8); state i gets the flag when run[e0.dst] >= 2.  Idempotent; every path that produces a host image
// ends here, so images written before the flag existed are upgraded on load.
static void finalize_runs(samd_static_t *s) {
    const int64_t n = s->n_states;
    std::vector<uint8_t> run((size_t)n + 1, 0);
    for (int64_t i = n - 1; i >= 0; i--) {
        const SamNode &nd = s->h_nodes[i];
        run[i] = (nd.e0_tok >= 0 && nd.e0_dst == i + 1) ? (uint8_t)std::min<int>(8, run[i + 1] + 1) : 0;
    }
    for (int64_t i = 0; i < n; i++) {
        SamNode &nd = s->h_nodes[i];
        nd.length &= ~SAMD_RUN;
        if (nd.e0_tok >= 0 && nd.e0_dst >= 0 && nd.e0_dst < n && run[nd.e0_dst] >= 2) nd.length |= SAMD_RUN;
    }
}

struct Ranked { int32_t tok, dst, key, order; };

// tables (dict-order edges) -> 64-byte node image
int layout(int32_t kind, int64_t n_states, const int32_t *link, const int32_t *length, const int32_t *aux,
           const int32_t *deg, const int32_t *edge_tok, const int32_t *edge_dst, const int32_t *text, int64_t n_text,
           samd_static_t **out) {
    if (n_states < 1 || n_states >= (1ll << 31)) { samd_set_error("bad state count"); return SAMD_E_INVALID; }
    samd_static_t *s = (samd_static_t *)calloc(1, sizeof(samd_static_t));
    if (!s) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    s->kind = kind; s->n_states = n_states; s->n_text = text ? n_text : 0;
    int64_t n_edges = 0, n_spill = 0;
    int32_t max_root_tok = -1;
    for (int64_t i = 0; i < n_states; i++) {
        if (deg[i] < 0) { free(s); samd_set_error("negative degree"); return SAMD_E_INVALID; }
        if (deg[i] > SAMD_INLINE_EDGES) n_spill += SAMD_SPILL_HEAD + samd_spill_slots(deg[i]);
        n_edges += deg[i];
    }
    if (n_edges > 0 && (!edge_tok || !edge_dst)) { free(s); samd_set_error("edge arrays missing"); return SAMD_E_INVALID; }
    for (int32_t j = 0; j < deg[0]; j++) max_root_tok = std::max(max_root_tok, edge_tok[j]);
    s->n_edges = n_edges; s->n_spill = n_spill; s->vocab = (int64_t)max_root_tok + 1;
    if (n_spill >= (1ll << 31)) { free(s); samd_set_error("spill region too large"); return SAMD_E_CAPACITY; }
    if (posix_memalign((void **)&s->h_nodes, 64, (size_t)n_states * sizeof(SamNode))) { free(s); return SAMD_E_CAPACITY; }
    s->h_root = (int32_t *)malloc(std::max<int64_t>(1, s->vocab) * sizeof(int32_t));
    s->h_spill = (SamEdge *)malloc(std::max<int64_t>(1, n_spill) * sizeof(SamEdge));
    s->h_text = (int32_t *)malloc(std::max<int64_t>(1, s->n_text) * sizeof(int32_t));
    if (!s->h_root || !s->h_spill || !s->h_text) { samd_static_free(s); samd_set_error("out of host memory for the automaton image"); return SAMD_E_CAPACITY; }
    if (text) memcpy(s->h_text, text, (size_t)n_text * sizeof(int32_t));
    for (int64_t t = 0; t < s->vocab; t++) s->h_root[t] = -1;

    std::vector<Ranked> row;
    int64_t ebase = 0, sp = 0;
    for (int64_t i = 0; i < n_states; i++) {
        const int32_t d = deg[i];
        SamNode &nd = s->h_nodes[i];
        if (length[i] < 0 || length[i] > SAMD_LEN_MASK) { samd_static_free(s); samd_set_error("state length out of range"); return SAMD_E_CAPACITY; }
        nd.link = link[i]; nd.length = length[i] | (d <= 1 ? SAMD_SINGLE : 0); nd.aux = aux[i]; nd.deg = d; nd.spill = -1; nd.reserved = 0;
        int32_t *words = reinterpret_cast<int32_t *>(&nd);
        for (int j = 0; j < SAMD_INLINE_EDGES; j++) { words[SAMD_EDGE_WORD(j)] = -1; words[SAMD_EDGE_WORD(j) + 1] = -1; }
        row.resize(d);
        for (int32_t j = 0; j < d; j++) {
            int32_t dst = edge_dst[ebase + j];
            if (dst < 0 || dst >= n_states) { samd_static_free(s); samd_set_error("edge target out of range"); return SAMD_E_INVALID; }
            row[j] = { edge_tok[ebase + j], dst, kind == SAMD_KIND_COUNT ? aux[dst] : 0, j };
        }
        if (i == 0) for (int32_t j = 0; j < d; j++) if (row[j].tok >= 0) s->h_root[row[j].tok] = row[j].dst;
        // ranks 0..7: stable descending by successor count (dict order breaks ties; for KIND_ENDPOS
        // every key is 0 so this is plain dict order)
        const int32_t top = std::min<int32_t>(d, SAMD_TOPK);
        std::partial_sort(row.begin(), row.begin() + top, row.end(), [](const Ranked &a, const Ranked &b) {
            return a.key != b.key ? a.key > b.key : a.order < b.order;
        });
        for (int32_t j = 0; j < std::min<int32_t>(d, SAMD_INLINE_EDGES); j++) { words[SAMD_EDGE_WORD(j)] = row[j].tok; words[SAMD_EDGE_WORD(j) + 1] = row[j].dst; }
        if (d > SAMD_INLINE_EDGES) {
            nd.spill = (int32_t)sp;
            for (int32_t j = 0; j < SAMD_SPILL_HEAD; j++) {
                int32_t r = SAMD_INLINE_EDGES + j;
                s->h_spill[sp + j] = r < d ? SamEdge{ row[r].tok, row[r].dst } : SamEdge{ -1, -1 };
            }
            const uint32_t m = samd_spill_slots(d);
            SamEdge *tab = s->h_spill + sp + SAMD_SPILL_HEAD;
            for (uint32_t j = 0; j < m; j++) tab[j] = SamEdge{ -1, -1 };
            for (int32_t j = SAMD_INLINE_EDGES; j < d; j++) {
                uint32_t h = samd_spill_hash(row[j].tok, m);
                while (tab[h].tok != -1) h = (h + 1) & (m - 1);
                tab[h] = SamEdge{ row[j].tok, row[j].dst };
            }
            sp += SAMD_SPILL_HEAD + m;
        }
        ebase += d;
    }
    finalize_runs(s);
    *out = s;
    return SAMD_OK;
}

struct FileHeader { char magic[8]; int64_t version, kind, n_states, n_edges, n_spill, vocab, n_text; };

}  // namespace

static bool image_is_sane(const samd_static_t *s, const char **why);

extern "C" {

int samd_static_build(const int32_t *h_tokens, const int64_t *h_doc_offsets, int64_t n_docs, int32_t eos_token,
                      int32_t kind, samd_static_t **out) {
    if (!out || (kind != SAMD_KIND_COUNT && kind != SAMD_KIND_ENDPOS) || n_docs < 0 || (n_docs > 0 && (!h_tokens || !h_doc_offsets))) {
        samd_set_error("samd_static_build: invalid argument"); return SAMD_E_INVALID;
    }
    try {
    Builder b(kind);
    for (int64_t d = 0; d < n_docs; d++) {
        int64_t lo = h_doc_offsets[d], hi = h_doc_offsets[d + 1];
        if (hi <= lo) { samd_set_error("samd_static_build: empty document %lld", (long long)d); return SAMD_E_INVALID; }
        b.add_document(h_tokens + lo, hi - lo, eos_token);
    }
    // flatten the per-state edge threads into state-major dict order
    std::vector<int32_t> etok(b.edge_tok_.size()), edst(b.edge_tok_.size());
    size_t k = 0;
    for (size_t s = 0; s < b.link_.size(); s++)
        for (int32_t e = b.first_[s]; e >= 0; e = b.edge_sibling_[e]) { etok[k] = b.edge_tok_[e]; edst[k] = b.edge_dst_[e]; k++; }
    return layout(kind, (int64_t)b.link_.size(), b.link_.data(), b.length_.data(), b.aux_.data(), b.deg_.data(),
                  etok.data(), edst.data(), kind == SAMD_KIND_ENDPOS ? b.text_.data() : nullptr, (int64_t)b.text_.size(), out);
    } catch (const std::exception &e) {                     // std::bad_alloc / length_error: never across the C ABI
        samd_set_error("samd_static_build: %s", e.what());
        return SAMD_E_CAPACITY;
    }
}

int samd_static_from_tables(int32_t kind, int64_t n_states, const int32_t *h_link, const int32_t *h_length,
                            const int32_t *h_aux, const int32_t *h_deg, const int32_t *h_edge_tok,
                            const int32_t *h_edge_dst, const int32_t *h_text, int64_t n_text, samd_static_t **out) {
    if (!out || !h_link || !h_length || !h_aux || !h_deg || (kind != SAMD_KIND_COUNT && kind != SAMD_KIND_ENDPOS)) {
        samd_set_error("samd_static_from_tables: invalid argument"); return SAMD_E_INVALID;
    }
    try {
        const int rc = layout(kind, n_states, h_link, h_length, h_aux, h_deg, h_edge_tok, h_edge_dst, h_text, n_text, out);
        if (rc != SAMD_OK) return rc;
        // tables come from outside the builder (a reference pickle, a converter): the same structural check a loaded image gets
        const char *why = "";
        if (!image_is_sane(*out, &why)) { samd_static_free(*out); *out = nullptr; samd_set_error("samd_static_from_tables: inconsistent tables (%s)", why); return SAMD_E_INVALID; }
        return SAMD_OK;
    } catch (const std::exception &e) {
        samd_set_error("samd_static_from_tables: %s", e.what());
        return SAMD_E_CAPACITY;
    }
}

void samd_static_free(samd_static_t *s) {
    if (!s) return;
    free(s->h_nodes); free(s->h_root); free(s->h_spill); free(s->h_text);
    if (!s->borrowed) {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_root) (void)hipFree(s->d_root);
        if (s->d_spill) (void)hipFree(s->d_spill);
        if (s->d_text) (void)hipFree(s->d_text);
    }
    if (s->d_chain) (void)hipFree(s->d_chain);
    if (s->d_root16) (void)hipFree(s->d_root16);
    if (s->d_d1hash) (void)hipFree(s->d_d1hash);
    if (s->d_rc_bits) (void)hipFree(s->d_rc_bits);
    if (s->d_topk_cnt) (void)hipFree(s->d_topk_cnt);
    if (s->d_ehash) (void)hipFree(s->d_ehash);
    if (s->d_hot) (void)hipFree(s->d_hot);
    if (s->d_blocks) (void)hipFree(s->d_blocks);
    free(s);
}

int samd_static_info(const samd_static_t *s, int64_t out[8]) {
    if (!s || !out) return SAMD_E_INVALID;
    out[0] = s->n_states; out[1] = s->n_edges; out[2] = s->n_spill; out[3] = s->vocab;
    out[4] = s->n_states * (int64_t)sizeof(SamNode) + s->vocab * 4 + s->n_spill * 8 + s->n_text * 4;
    out[5] = s->kind; out[6] = s->n_text; out[7] = s->uploaded;
    return SAMD_OK;
}

int samd_static_derived_info(const samd_static_t *s, int64_t out[6]) {
    if (!s || !out) return SAMD_E_INVALID;
    out[4] = s->d_ehash ? s->n_ehash * 16 : 0;
    out[5] = s->d_ehash ? s->n_ehash : 0;
    out[0] = s->d_chain ? s->n_states * 16 : 0;
    out[1] = s->d_d1hash ? s->n_d1hash * 16 + s->vocab * 16 + ((s->vocab + 31) / 32) * 4 : 0;
    out[2] = s->d_topk_cnt ? s->n_states * (int64_t)SAMD_TOPK * 4 : 0;
    out[3] = s->d_d1hash ? s->n_d1hash : 0;
    return SAMD_OK;
}

int samd_static_edge_blocks_info(const samd_static_t *s, int64_t out[4]) {
    if (!s || !out) return SAMD_E_INVALID;
    const bool on = s->d_hot && s->d_blocks;
    out[0] = on ? s->n_states * 16 : 0;
    out[1] = on ? (s->n_block_slots ? s->n_block_slots : 1) * 16 : 0;
    out[2] = on ? s->n_block_slots : 0;
    out[3] = on ? s->n_block_states : 0;
    return SAMD_OK;
}

int samd_static_export(const samd_static_t *s, int32_t *h_link, int32_t *h_length, int32_t *h_aux, int32_t *h_deg,
                       int32_t *h_edge_tok, int32_t *h_edge_dst) {
    if (!s || !s->h_nodes) { samd_set_error("samd_static_export: no host image"); return SAMD_E_INVALID; }
    int64_t k = 0;
    for (int64_t i = 0; i < s->n_states; i++) {
        const SamNode &nd = s->h_nodes[i];
        if (h_link) h_link[i] = nd.link;
        if (h_length) h_length[i] = nd.length & SAMD_LEN_MASK;
        if (h_aux) h_aux[i] = nd.aux;
        if (h_deg) h_deg[i] = nd.deg;
        if (!h_edge_tok || !h_edge_dst) continue;
        const int32_t d = nd.deg;
        const int32_t *words = reinterpret_cast<const int32_t *>(&nd);
        for (int32_t j = 0; j < std::min<int32_t>(d, SAMD_INLINE_EDGES); j++) { h_edge_tok[k] = words[SAMD_EDGE_WORD(j)]; h_edge_dst[k] = words[SAMD_EDGE_WORD(j) + 1]; k++; }
        if (d > SAMD_INLINE_EDGES) {
            const SamEdge *sp = s->h_spill + nd.spill;
            int32_t nhead = std::min<int32_t>(d, SAMD_TOPK) - SAMD_INLINE_EDGES;
            for (int32_t j = 0; j < nhead; j++) { h_edge_tok[k] = sp[j].tok; h_edge_dst[k] = sp[j].dst; k++; }
            // the hashed remainder, ascending by token (the order the export contract promises)
            std::vector<SamEdge> rest;
            const uint32_t m = samd_spill_slots(d);
            for (uint32_t j = 0; j < m; j++) {
                const SamEdge &e = sp[SAMD_SPILL_HEAD + j];
                if (e.tok == -1) continue;
                bool in_head = false;
                for (int32_t h = 0; h < nhead; h++) in_head |= (sp[h].tok == e.tok);
                if (!in_head) rest.push_back(e);
            }
            std::sort(rest.begin(), rest.end(), [](const SamEdge &a, const SamEdge &b) { return a.tok < b.tok; });
            for (const SamEdge &e : rest) { h_edge_tok[k] = e.tok; h_edge_dst[k] = e.dst; k++; }
        }
    }
    return SAMD_OK;
}

int samd_static_save(const samd_static_t *s, const char *path) {
    if (!s || !path || !s->h_nodes) return SAMD_E_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) { samd_set_error("cannot open %s for writing", path); return SAMD_E_IO; }
    FileHeader h; memset(&h, 0, sizeof(h)); memcpy(h.magic, "SAMDHIP1", 8);
    h.version = SAMD_ABI_VERSION; h.kind = s->kind; h.n_states = s->n_states; h.n_edges = s->n_edges;
    h.n_spill = s->n_spill; h.vocab = s->vocab; h.n_text = s->n_text;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fwrite(s->h_nodes, sizeof(SamNode), (size_t)s->n_states, f) == (size_t)s->n_states;
    ok = ok && fwrite(s->h_root, 4, (size_t)s->vocab, f) == (size_t)s->vocab;
    ok = ok && fwrite(s->h_spill, 8, (size_t)s->n_spill, f) == (size_t)s->n_spill;
    ok = ok && fwrite(s->h_text, 4, (size_t)s->n_text, f) == (size_t)s->n_text;
    ok = (fclose(f) == 0) && ok;
    if (!ok) { samd_set_error("short write to %s", path); return SAMD_E_IO; }
    return SAMD_OK;
}


// Structural check of a host image that came from outside the builder (a file, another rank): every index a kernel
// will follow must stay inside the image, otherwise a damaged file turns into a wild device read.
static bool image_is_sane(const samd_static_t *s, const char **why) {
    const int64_t n = s->n_states;
    int64_t n_edges = 0;
    auto dst_ok = [&](int32_t tok, int32_t dst) { return tok < 0 ? true : (dst >= 0 && dst < n); };
    if (s->kind != SAMD_KIND_COUNT && s->kind != SAMD_KIND_ENDPOS) { *why = "unknown automaton kind"; return false; }
    if (s->h_nodes[0].link != -1 || (s->h_nodes[0].length & SAMD_LEN_MASK) != 0) { *why = "state 0 is not a root"; return false; }
    for (int64_t i = 0; i < n; i++) {
        const SamNode &nd = s->h_nodes[i];
        // transfer_state climbs `index = states[index].link` until the root (static_sam.py:99-101): every non-root link must name a state of
        // strictly smaller length, or a damaged image becomes a walk that never ends (or leaves the image through link -1) on the device
        if (i > 0 && (nd.link < 0 || nd.link >= n)) { *why = "suffix link out of range"; return false; }
        if (i > 0 && (s->h_nodes[nd.link].length & SAMD_LEN_MASK) >= (nd.length & SAMD_LEN_MASK)) { *why = "suffix link does not shorten the match"; return false; }
        if (nd.deg < 0) { *why = "negative degree"; return false; }
        if (!dst_ok(nd.e0_tok, nd.e0_dst) || !dst_ok(nd.e1_tok, nd.e1_dst) || !dst_ok(nd.e2_tok, nd.e2_dst) ||
            !dst_ok(nd.e3_tok, nd.e3_dst) || !dst_ok(nd.e4_tok, nd.e4_dst)) { *why = "edge target out of range"; return false; }
        if (nd.deg > SAMD_INLINE_EDGES) {
            if ((int64_t)nd.deg - SAMD_INLINE_EDGES > s->n_spill) { *why = "degree exceeds the spill region"; return false; }
            int64_t slots = (int64_t)SAMD_SPILL_HEAD + samd_spill_slots(nd.deg);
            if (nd.spill < 0 || (int64_t)nd.spill + slots > s->n_spill) { *why = "spill block out of range"; return false; }
            for (int64_t k = 0; k < slots; k++) {
                const SamEdge &e = s->h_spill[nd.spill + k];
                if (!dst_ok(e.tok, e.dst)) { *why = "spill edge target out of range"; return false; }
            }
            // the block must hold exactly the edges `deg` promises (ranks 5..7 up front, repeated in the hashed part): samd_static_export
            // writes what it finds there into arrays the caller sized from the degrees
            const SamEdge *sp = s->h_spill + nd.spill;
            const int32_t nhead = std::min<int32_t>(nd.deg, SAMD_TOPK) - SAMD_INLINE_EDGES;
            int64_t rest = 0;
            for (int64_t k = SAMD_SPILL_HEAD; k < slots; k++) {
                if (sp[k].tok == -1) continue;
                bool in_head = false;
                for (int32_t h = 0; h < nhead; h++) in_head |= (sp[h].tok == sp[k].tok);
                rest += !in_head;
            }
            if (SAMD_INLINE_EDGES + nhead + rest != nd.deg) { *why = "spill block does not hold the state's degree"; return false; }
        }
        n_edges += nd.deg;
        if (s->kind == SAMD_KIND_ENDPOS && (nd.aux < 0 || nd.aux >= std::max<int64_t>(1, s->n_text))) {
            *why = "end position outside the text"; return false;
        }
    }
    for (int64_t t = 0; t < s->vocab; t++)
        if (s->h_root[t] < -1 || s->h_root[t] >= n) { *why = "root table entry out of range"; return false; }
    if (n_edges != s->n_edges) { *why = "edge count differs from the sum of the degrees"; return false; }     // (callers size export arrays from it)
    return true;
}

int samd_static_load(const char *path, samd_static_t **out) {
    if (!path || !out) return SAMD_E_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) { samd_set_error("cannot open %s", path); return SAMD_E_IO; }
    FileHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SAMDHIP1", 8) != 0 || h.version != SAMD_ABI_VERSION ||
        h.n_states < 1 || h.n_spill < 0 || h.vocab < 0 || h.n_text < 0) {
        fclose(f); samd_set_error("%s: not a SAMDHIP1 image", path); return SAMD_E_IO;
    }
    // the header's sizes against the file's length BEFORE anything is allocated from them: a damaged header must not become a
    // multi-terabyte malloc (or a size_t overflow) -- every count is first bounded by what a file of this length could hold
    long file_len = -1;
    if (fseek(f, 0, SEEK_END) == 0) file_len = ftell(f);
    const int64_t body = (int64_t)file_len - (int64_t)sizeof(FileHeader);
    if (file_len < 0 || fseek(f, (long)sizeof(FileHeader), SEEK_SET) != 0 || body < 0 || h.n_states >= (1ll << 31) || h.n_edges < 0 ||
        h.n_states > body / (int64_t)sizeof(SamNode) || h.vocab > body / 4 || h.n_spill > body / 8 || h.n_text > body / 4 ||
        h.n_states * (int64_t)sizeof(SamNode) + h.vocab * 4 + h.n_spill * 8 + h.n_text * 4 != body ||
        (h.kind != SAMD_KIND_COUNT && h.kind != SAMD_KIND_ENDPOS)) {
        fclose(f); samd_set_error("%s: truncated or damaged image (header and file length disagree)", path); return SAMD_E_IO;
    }
    samd_static_t *s = (samd_static_t *)calloc(1, sizeof(samd_static_t));
    if (!s) { fclose(f); samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    s->kind = (int32_t)h.kind; s->n_states = h.n_states; s->n_edges = h.n_edges; s->n_spill = h.n_spill;
    s->vocab = h.vocab; s->n_text = h.n_text;
    bool ok = posix_memalign((void **)&s->h_nodes, 64, (size_t)h.n_states * sizeof(SamNode)) == 0;
    s->h_root = (int32_t *)malloc(std::max<int64_t>(1, h.vocab) * 4);
    s->h_spill = (SamEdge *)malloc(std::max<int64_t>(1, h.n_spill) * 8);
    s->h_text = (int32_t *)malloc(std::max<int64_t>(1, h.n_text) * 4);
    ok = ok && s->h_root && s->h_spill && s->h_text;
    ok = ok && fread(s->h_nodes, sizeof(SamNode), (size_t)h.n_states, f) == (size_t)h.n_states;
    ok = ok && fread(s->h_root, 4, (size_t)h.vocab, f) == (size_t)h.vocab;
    ok = ok && fread(s->h_spill, 8, (size_t)h.n_spill, f) == (size_t)h.n_spill;
    ok = ok && fread(s->h_text, 4, (size_t)h.n_text, f) == (size_t)h.n_text;
    fclose(f);
    if (!ok) { samd_static_free(s); samd_set_error("%s: truncated image", path); return SAMD_E_IO; }
    const char *why = "";
    if (!image_is_sane(s, &why)) { samd_static_free(s); samd_set_error("%s: damaged image (%s)", path, why); return SAMD_E_IO; }
    finalize_runs(s);
    *out = s;
    return SAMD_OK;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { samd_set_error("%s: %s", #x, hipGetErrorString(e_)); return SAMD_E_HIP; } } while (0)

static int alloc_device_image(samd_static_t *s) {
    HIPCHK(hipMalloc((void **)&s->d_nodes, (size_t)s->n_states * sizeof(SamNode)));
    HIPCHK(hipMalloc((void **)&s->d_root, std::max<int64_t>(1, s->vocab) * 4));
    HIPCHK(hipMalloc((void **)&s->d_spill, std::max<int64_t>(1, s->n_spill) * 8));
    HIPCHK(hipMalloc((void **)&s->d_text, std::max<int64_t>(1, s->n_text) * 4));
    return SAMD_OK;
}

int samd_static_upload(samd_static_t *s) {
    if (!s) return SAMD_E_INVALID;
    if (s->uploaded) return SAMD_OK;
    if (!s->h_nodes) { samd_set_error("samd_static_upload: no host image"); return SAMD_E_INVALID; }
    int rc = alloc_device_image(s);
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->d_nodes, s->h_nodes, (size_t)s->n_states * sizeof(SamNode), hipMemcpyHostToDevice));
    if (s->vocab) HIPCHK(hipMemcpy(s->d_root, s->h_root, (size_t)s->vocab * 4, hipMemcpyHostToDevice));
    if (s->n_spill) HIPCHK(hipMemcpy(s->d_spill, s->h_spill, (size_t)s->n_spill * 8, hipMemcpyHostToDevice));
    if (s->n_text) HIPCHK(hipMemcpy(s->d_text, s->h_text, (size_t)s->n_text * 4, hipMemcpyHostToDevice));
    s->uploaded = 1;
    return samd_static_derive_chain(s, nullptr);
}

int samd_static_device_image(const samd_static_t *s, void *out_ptrs[4], int64_t out_bytes[4]) {
    if (!s || !s->uploaded) { samd_set_error("samd_static_device_image: not uploaded"); return SAMD_E_INVALID; }
    out_ptrs[0] = s->d_nodes; out_bytes[0] = s->n_states * (int64_t)sizeof(SamNode);
    out_ptrs[1] = s->d_root; out_bytes[1] = s->vocab * 4;
    out_ptrs[2] = s->d_spill; out_bytes[2] = s->n_spill * 8;
    out_ptrs[3] = s->d_text; out_bytes[3] = s->n_text * 4;
    return SAMD_OK;
}

int samd_static_host_image(const samd_static_t *s, void *out_ptrs[4], int64_t out_bytes[4]) {
    if (!s || !s->h_nodes || !out_ptrs || !out_bytes) { samd_set_error("samd_static_host_image: no host image"); return SAMD_E_INVALID; }
    out_ptrs[0] = s->h_nodes; out_bytes[0] = s->n_states * (int64_t)sizeof(SamNode);
    out_ptrs[1] = s->h_root; out_bytes[1] = s->vocab * 4;
    out_ptrs[2] = s->h_spill; out_bytes[2] = s->n_spill * 8;
    out_ptrs[3] = s->h_text; out_bytes[3] = s->n_text * 4;
    return SAMD_OK;
}

static samd_static_t *shell_from_info(const int64_t info[8]) {
    samd_static_t *s = (samd_static_t *)calloc(1, sizeof(samd_static_t));
    if (!s) return nullptr;
    s->n_states = info[0]; s->n_edges = info[1]; s->n_spill = info[2]; s->vocab = info[3];
    s->kind = (int32_t)info[5]; s->n_text = info[6];
    return s;
}

int samd_static_from_host_image(const int64_t info[8], const void *const h_ptrs[4], samd_static_t **out) {
    if (!info || !h_ptrs || !out || info[0] < 1 || !h_ptrs[0]) { samd_set_error("samd_static_from_host_image: invalid argument"); return SAMD_E_INVALID; }
    samd_static_t *s = shell_from_info(info);
    if (!s) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    if (posix_memalign((void **)&s->h_nodes, 64, (size_t)s->n_states * sizeof(SamNode))) { free(s); return SAMD_E_CAPACITY; }
    s->h_root = (int32_t *)malloc(std::max<int64_t>(1, s->vocab) * 4);
    s->h_spill = (SamEdge *)malloc(std::max<int64_t>(1, s->n_spill) * 8);
    s->h_text = (int32_t *)malloc(std::max<int64_t>(1, s->n_text) * 4);
    if (!s->h_root || !s->h_spill || !s->h_text) { samd_static_free(s); samd_set_error("out of host memory for the automaton image"); return SAMD_E_CAPACITY; }
    memcpy(s->h_nodes, h_ptrs[0], (size_t)s->n_states * sizeof(SamNode));
    if (s->vocab && h_ptrs[1]) memcpy(s->h_root, h_ptrs[1], (size_t)s->vocab * 4);
    if (s->n_spill && h_ptrs[2]) memcpy(s->h_spill, h_ptrs[2], (size_t)s->n_spill * 8);
    if (s->n_text && h_ptrs[3]) memcpy(s->h_text, h_ptrs[3], (size_t)s->n_text * 4);
    const char *why = "";
    if (!image_is_sane(s, &why)) { samd_static_free(s); samd_set_error("samd_static_from_host_image: damaged image (%s)", why); return SAMD_E_INVALID; }
    finalize_runs(s);
    *out = s;
    return SAMD_OK;
}

int samd_static_adopt_device(const int64_t info[8], void *const d_ptrs[4], samd_static_t **out) {
    if (!info || !d_ptrs || !out || info[0] < 1 || !d_ptrs[0]) { samd_set_error("samd_static_adopt_device: invalid argument"); return SAMD_E_INVALID; }
    samd_static_t *s = shell_from_info(info);
    if (!s) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    s->d_nodes = (SamNode *)d_ptrs[0]; s->d_root = (int32_t *)d_ptrs[1]; s->d_spill = (SamEdge *)d_ptrs[2]; s->d_text = (int32_t *)d_ptrs[3];
    s->uploaded = 1; s->borrowed = 1;
    const int rc = samd_static_derive_chain(s, nullptr);      // the caller's four regions are filled before they are adopted
    if (rc) { samd_static_free(s); return rc; }
    *out = s;
    return SAMD_OK;
}

int samd_static_alloc_like(const int64_t info[8], samd_static_t **out) {
    if (!info || !out || info[0] < 1) return SAMD_E_INVALID;
    samd_static_t *s = (samd_static_t *)calloc(1, sizeof(samd_static_t));
    if (!s) { samd_set_error("out of host memory"); return SAMD_E_CAPACITY; }
    s->n_states = info[0]; s->n_edges = info[1]; s->n_spill = info[2]; s->vocab = info[3];
    s->kind = (int32_t)info[5]; s->n_text = info[6];
    int rc = alloc_device_image(s);
    if (rc) { samd_static_free(s); return rc; }
    s->uploaded = 1;
    *out = s;
    return SAMD_OK;
}

}  // extern "C"
