"""ctypes view of oracle/sam_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Class and method names follow the reference (samd_sam_only/sam, samd/sam,
samd_sam_only/draft.py, samd_sam_only/utils.py) so that tests read like reference usage.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)


def build(force=False):
    if os.environ.get("SAM_ORACLE_LIB"):                   # a sanitizer build of the same source (scripts/asan_cpu.sh)
        return os.environ["SAM_ORACLE_LIB"]
    so = os.path.join(_HERE, "libsam_oracle.so")
    src = os.path.join(_HERE, "sam_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsam_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.osam_new.restype = C.c_void_p
        L.osam_new.argtypes = [C.c_int32]
        for name in ("osam_free", "osam_reset_all", "osam_reset_cursor", "osam_init_topk"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.osam_transfer_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, i32p, i32p]
        L.osam_lookup.argtypes = [C.c_void_p, C.c_int32, i32p, i32p]
        L.osam_transfer_tokens.argtypes = [C.c_void_p, i32p, C.c_int64]
        L.osam_add_tokens.argtypes = [C.c_void_p, i32p, C.c_int64]
        L.osam_add_batch.argtypes = [C.c_void_p, i32p, i64p, C.c_int64, C.c_int32]
        for name in ("osam_num_states", "osam_num_edges", "osam_text_len"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int64
        for name in ("osam_last", "osam_max_length"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int32
        L.osam_cursor.argtypes = [C.c_void_p, i32p, i32p]
        L.osam_set_cursor.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.osam_export_states.argtypes = [C.c_void_p, i32p, i32p, i32p, i32p]
        L.osam_export_edges.argtypes = [C.c_void_p, i32p, i32p]
        L.osam_export_text.argtypes = [C.c_void_p, i32p]
        L.osam_export_topk.argtypes = [C.c_void_p, i32p, i32p, i32p]
        L.osam_gen_draft_seq.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, i32p]
        L.osam_gen_draft_seq.restype = C.c_int32
        L.osam_to_anc.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.osam_to_anc.restype = C.c_int32
        L.osam_gen_draft_fixed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p]
        L.osam_gen_draft_tree.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                          C.c_int32, i32p, i32p]
        L.osam_gen_draft_tree.restype = C.c_int32
        L.o_gen_buffers.argtypes = [i32p, C.c_int32, i64p, u8p, i64p, i32p, i32p]
        L.o_tr_gen_buffers.argtypes = [i32p, i32p, C.c_int32, i32p, i64p, u8p, i64p, i32p, i32p]
        L.o_argmax_rows.argtypes = [f32p, C.c_int64, C.c_int64, i32p]
        L.o_candidates.argtypes = [i32p, C.c_int32, i64p, C.c_int32, C.c_int32, i32p]
        L.o_eval_posterior.argtypes = [i32p, i32p, C.c_int32, i64p, C.c_int32, C.c_int32, i32p, i32p, i32p]
        L.o_tr_update.argtypes = [i32p, u8p, i32p, i32p, C.c_int32]
        L.o_tr_gen_draft.argtypes = [i32p, u8p, i32p, i32p, C.c_int32, C.c_int32, i32p]
        L.o_topk8_rows.argtypes = [f32p, C.c_int64, C.c_int64, i32p]
        L.o_draft_lookup_so.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                        C.c_int32, i32p, i32p, i32p]
        L.o_draft_lookup_so.restype = C.c_int32
        L.o_draft_update.argtypes = [C.c_void_p, C.c_void_p, i32p, C.c_int32]
        _LIB = L
    return _LIB


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(i64p)


class _SAM:
    """Common part of DynSAM / StaticSAM (both variants)."""
    KIND = 1

    def __init__(self):
        self._h = C.c_void_p(lib().osam_new(self.KIND))

    def __del__(self):
        try:
            lib().osam_free(self._h)
        except Exception:
            pass

    # reference: transfer_state / lookup / transfer_tokens / add_tokens
    def transfer_state(self, index, length, token):
        oi, ol = C.c_int32(), C.c_int32()
        lib().osam_transfer_state(self._h, index, length, token, C.byref(oi), C.byref(ol))
        return oi.value, ol.value

    def lookup(self, token):
        oi, ol = C.c_int32(), C.c_int32()
        lib().osam_lookup(self._h, token, C.byref(oi), C.byref(ol))
        return oi.value, ol.value

    def transfer_tokens(self, tokens):
        a, p = _i32(tokens)
        lib().osam_transfer_tokens(self._h, p, len(a))

    def add_tokens(self, tokens):
        a, p = _i32(tokens)
        lib().osam_add_tokens(self._h, p, len(a))

    def add_batch_tokens(self, batch_tokens, eos_token, verbose=False):
        flat = np.concatenate([np.asarray(t, dtype=np.int32) for t in batch_tokens])
        off = np.zeros(len(batch_tokens) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(t) for t in batch_tokens])
        fa, fp = _i32(flat)
        oa, op = _i64(off)
        lib().osam_add_batch(self._h, fp, op, len(batch_tokens), eos_token)

    @property
    def cur_index(self):
        return self.cursor()[0]

    @property
    def cur_length(self):
        return self.cursor()[1]

    def cursor(self):
        oi, ol = C.c_int32(), C.c_int32()
        lib().osam_cursor(self._h, C.byref(oi), C.byref(ol))
        return oi.value, ol.value

    def set_cursor(self, index, length):
        lib().osam_set_cursor(self._h, index, length)

    @property
    def num_states(self):
        return lib().osam_num_states(self._h)

    @property
    def last(self):
        return lib().osam_last(self._h)

    @property
    def max_length(self):
        return lib().osam_max_length(self._h)

    def export(self):
        """-> dict(link, length, aux, deg, edge_tok, edge_dst[, text]); edges state-major, dict order."""
        n = self.num_states
        link, length, aux, deg = (np.empty(n, np.int32) for _ in range(4))
        lib().osam_export_states(self._h, *(x.ctypes.data_as(i32p) for x in (link, length, aux, deg)))
        ne = lib().osam_num_edges(self._h)
        et, ed = np.empty(ne, np.int32), np.empty(ne, np.int32)
        lib().osam_export_edges(self._h, et.ctypes.data_as(i32p), ed.ctypes.data_as(i32p))
        out = dict(link=link, length=length, aux=aux, deg=deg, edge_tok=et, edge_dst=ed)
        if self.KIND == 1:
            t = np.empty(lib().osam_text_len(self._h), np.int32)
            lib().osam_export_text(self._h, t.ctypes.data_as(i32p))
            out["text"] = t
        return out


class DynSAM(_SAM):
    """samd_sam_only/sam/dyn_sam.py DynSAM (also the full variant's, via n_predicts)."""
    KIND = 1

    def __init__(self, max_predicts=40, alpha=4.0, device="cpu", n_predicts=40):
        super().__init__()
        self.max_predicts, self.alpha, self.device, self.n_predicts = max_predicts, alpha, device, n_predicts

    def reset(self):
        lib().osam_reset_all(self._h)

    def gen_draft(self, index, match_length, start_token):
        """sam_only: variable-length sequence (SO/sam/dyn_sam.py:116-121)."""
        out = np.empty(max(self.max_predicts, 1) + 1, np.int32)
        m = lib().osam_gen_draft_seq(self._h, index, match_length, start_token, self.max_predicts,
                                     float(self.alpha), out.ctypes.data_as(i32p))
        return out[:m].tolist()

    def to_anc(self, index):
        return lib().osam_to_anc(self._h, index, self.n_predicts)

    def gen_draft_fixed(self, index, start_token):
        """full variant: to_anc + fixed n_predicts, zero padded (S/sam/dyn_sam.py:107-113)."""
        out = np.empty(self.n_predicts, np.int32)
        lib().osam_gen_draft_fixed(self._h, index, start_token, self.n_predicts, 1, out.ctypes.data_as(i32p))
        return out.tolist()


class StaticSAM(_SAM):
    """samd_sam_only/sam/static_sam.py StaticSAM (counts + top-k + tree draft)."""
    KIND = 0

    def __init__(self, max_predicts=40, alpha=4.0, K=8, device="cpu"):
        super().__init__()
        self.max_predicts, self.alpha, self.K, self.device = max_predicts, alpha, K, device

    @staticmethod
    def build(batch_tokens, eos_token, verbose=False):
        sam = StaticSAM()
        sam.add_batch_tokens(batch_tokens, eos_token, verbose)
        sam.init_topk_next()
        return sam

    def reset(self):
        lib().osam_reset_cursor(self._h)

    def init_topk_next(self):
        lib().osam_init_topk(self._h)

    def export_topk(self):
        n = self.num_states
        tok, dst, cnt = np.empty((n, 8), np.int32), np.empty((n, 8), np.int32), np.empty(n, np.int32)
        lib().osam_export_topk(self._h, *(x.ctypes.data_as(i32p) for x in (tok, dst, cnt)))
        return tok, dst, cnt

    def gen_draft_tree(self, index, match_length, start_token):
        """-> (tree tokens, anc_tree)  (SO/sam/static_sam.py:182-215, before gen_buffers)."""
        cap = max(self.max_predicts, 1) + 1
        tree, anc = np.empty(cap, np.int32), np.empty(cap, np.int32)
        m = lib().osam_gen_draft_tree(self._h, index, match_length, start_token, self.max_predicts,
                                      float(self.alpha), self.K, tree.ctypes.data_as(i32p), anc.ctypes.data_as(i32p))
        return tree[:m].tolist(), anc[:m].tolist()

    def gen_draft(self, index, match_length, start_token):
        tree, anc = self.gen_draft_tree(index, match_length, start_token)
        return tree, gen_buffers(anc)


class StaticSAMFull(_SAM):
    """samd/sam/static_sam.py StaticSAM (min_endpos + input_ids, fixed-length sequence draft)."""
    KIND = 1

    def __init__(self, n_predicts=40):
        super().__init__()
        self.n_predicts = n_predicts

    @staticmethod
    def build(batch_tokens, eos_token, verbose=False):
        sam = StaticSAMFull()
        sam.add_batch_tokens(batch_tokens, eos_token, verbose)
        return sam

    def reset(self):
        lib().osam_reset_cursor(self._h)

    def gen_draft(self, index, start_token):
        out = np.empty(self.n_predicts, np.int32)
        lib().osam_gen_draft_fixed(self._h, index, start_token, self.n_predicts, 0, out.ctypes.data_as(i32p))
        return out.tolist()


def gen_buffers(anc_tree):
    """SO/sam/static_sam.py:148-180 -> dict of numpy arrays shaped like the reference tensors."""
    a, p = _i32(anc_tree)
    n = len(a)
    pos = np.empty(n, np.int64)
    mask = np.empty((n, n), np.uint8)
    ret = np.empty(n * n, np.int64)
    nl, md = C.c_int32(), C.c_int32()
    lib().o_gen_buffers(p, n, pos.ctypes.data_as(i64p), mask.ctypes.data_as(u8p), ret.ctypes.data_as(i64p),
                        C.byref(nl), C.byref(md))
    return {
        "tree_attn_mask": mask.astype(bool).reshape(1, 1, n, n),
        "tree_position_ids": pos.reshape(1, n),
        "tree_retrieve_indices": ret[: nl.value * md.value].reshape(nl.value, md.value).copy(),
    }


def _flatten_tree(tree):
    off = np.zeros(len(tree) + 1, np.int32)
    off[1:] = np.cumsum([len(c) for c in tree])
    ch = np.asarray([c for cs in tree for c in cs], dtype=np.int32)
    if ch.size == 0:
        ch = np.zeros(1, np.int32)
    return off, ch


def tr_gen_buffers(tree):
    """S/tree_model/token_recycle/utils.py:37-99 for a child-list tree."""
    off, ch = _flatten_tree(tree)
    n = len(tree)
    anc = np.empty(n, np.int32)
    pos = np.empty(n, np.int64)
    mask = np.empty((n, n), np.uint8)
    ret = np.empty(n * n, np.int64)
    nl, md = C.c_int32(), C.c_int32()
    lib().o_tr_gen_buffers(off.ctypes.data_as(i32p), ch.ctypes.data_as(i32p), n, anc.ctypes.data_as(i32p),
                           pos.ctypes.data_as(i64p), mask.ctypes.data_as(u8p), ret.ctypes.data_as(i64p),
                           C.byref(nl), C.byref(md))
    return {
        "anc_tree": anc,
        "tree_attn_mask": mask.astype(np.float32).reshape(1, 1, n, n),
        "tree_position_ids": pos.reshape(1, n),
        "tree_retrieve_indices": ret[: nl.value * md.value].reshape(nl.value, md.value).copy(),
    }


def argmax_rows(logits):
    x = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.empty(x.shape[0], np.int32)
    lib().o_argmax_rows(x.ctypes.data_as(f32p), x.shape[0], x.shape[1], out.ctypes.data_as(i32p))
    return out


def topk8_rows(logits):
    x = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.empty((x.shape[0], 8), np.int32)
    lib().o_topk8_rows(x.ctypes.data_as(f32p), x.shape[0], x.shape[1], out.ctypes.data_as(i32p))
    return out


def candidates(tokens, retrieve):
    """(tokens + [0])[retrieve]  (SO/utils.py:94-97)."""
    t, tp = _i32(tokens)
    r, rp = _i64(retrieve)
    out = np.empty(r.shape, np.int32)
    lib().o_candidates(tp, len(t), rp, r.shape[0], r.shape[1], out.ctypes.data_as(i32p))
    return out


def eval_posterior(node_argmax, tokens, retrieve=None):
    """Greedy eval_posterior on per-node arg-max tokens (SO/utils.py:127-141).
    -> (best_candidate, accept_length, next_node)."""
    am, amp = _i32(node_argmax)
    t, tp = _i32(tokens)
    b, a, nn = C.c_int32(), C.c_int32(), C.c_int32()
    if retrieve is None:
        lib().o_eval_posterior(amp, tp, len(t), None, 1, len(t), C.byref(b), C.byref(a), C.byref(nn))
    else:
        r, rp = _i64(retrieve)
        lib().o_eval_posterior(amp, tp, len(t), rp, r.shape[0], r.shape[1], C.byref(b), C.byref(a), C.byref(nn))
    return b.value, a.value, nn.value


class TokenRecycle:
    """S/tree_model/token_recycle/token_recycle.py as a dense [V, 8] table."""

    def __init__(self, tree, vocab):
        self.tree = tree
        self.off, self.ch = _flatten_tree(tree)
        self.table = np.zeros((vocab, 8), np.int32)
        self.present = np.zeros(vocab, np.uint8)

    def reset(self):
        pass

    def update(self, tree_tokens, topk):
        t, tp = _i32(tree_tokens)
        k, kp = _i32(topk)
        lib().o_tr_update(self.table.ctypes.data_as(i32p), self.present.ctypes.data_as(u8p), tp, kp, len(t))

    def gen_draft(self, start_token):
        out = np.empty(len(self.tree), np.int32)
        lib().o_tr_gen_draft(self.table.ctypes.data_as(i32p), self.present.ctypes.data_as(u8p),
                             self.off.ctypes.data_as(i32p), self.ch.ctypes.data_as(i32p), len(self.tree),
                             start_token, out.ctypes.data_as(i32p))
        return out.tolist()


class DraftModel:
    """samd_sam_only/draft.py DraftModel on the CPU oracle."""

    def __init__(self, max_predicts=60, alpha=4.0, K=8, len_bias=5, sam_dyn=None, sam_static=None):
        self.sam_dyn = sam_dyn if sam_dyn is not None else DynSAM(max_predicts, alpha)
        self.sam_static = sam_static if sam_static is not None else StaticSAM(max_predicts, alpha, K)
        if sam_static is None:
            self.sam_static.init_topk_next()
        for s in (self.sam_dyn, self.sam_static):
            s.max_predicts, s.alpha = max_predicts, alpha
        self.sam_static.K = K
        self.max_predicts, self.alpha, self.K, self.len_bias = max_predicts, alpha, K, len_bias

    def reset(self):
        self.sam_dyn.reset()
        self.sam_static.reset()

    def lookup_raw(self, start_token):
        """-> (type 0|1, tokens, anc_tree)."""
        cap = max(self.max_predicts, 1) + 1
        tok, anc = np.empty(cap, np.int32), np.empty(cap, np.int32)
        n = C.c_int32()
        ty = lib().o_draft_lookup_so(self.sam_dyn._h, self.sam_static._h, start_token, self.max_predicts,
                                     float(self.alpha), self.K, self.len_bias, tok.ctypes.data_as(i32p),
                                     anc.ctypes.data_as(i32p), C.byref(n))
        return ty, tok[: n.value].tolist(), anc[: n.value].tolist()

    def lookup(self, start_token):
        ty, tok, anc = self.lookup_raw(start_token)
        if ty == 0:
            return "sequence", tok, {"seq_position_ids": np.arange(len(tok), dtype=np.int64).reshape(1, -1)}
        return "tree", tok, gen_buffers(anc)

    def update(self, tokens):
        a, p = _i32(tokens)
        lib().o_draft_update(self.sam_dyn._h, self.sam_static._h, p, len(a))
