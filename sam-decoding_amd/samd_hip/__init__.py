"""samd_hip -- ctypes binding of libsamd_hip.so (include/samd_hip.h), the gfx950 draft+verify hot path.

There is NO CPU fallback: every op raises if the shared library is missing or if no MI355X is
visible.  torch is used only for device memory, streams and dtypes.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAMD_HIP_LIB: another build of the SAME library (scripts/asan_cpu.sh points the CPU tests at a host-sanitizer build); it must
# export every entry point like the default one, and a missing file fails just as loudly
LIB_PATH = os.environ.get("SAMD_HIP_LIB") or os.path.join(_HERE, "libsamd_hip.so")
MAX_DRAFT = 128             # nodes of a draft (include/samd_hip.h SAMD_MAX_DRAFT; 64 until round 5)
TILE_ROWS = 64              # rows of one verify tile: the streaming GEMM's largest row tile, one u64 mask word, a prefill chunk
TOPK = 8
# report block layout (include/samd_hip.h SAMD_REP_*)
REP_DMETA, REP_VERDICT, REP_TOKENS, REP_KVINDEX, REP_COUNTERS, REP_META, REPORT_INTS = 0, 16, 24, 152, 280, 288, 304
KIND_COUNT, KIND_ENDPOS = 0, 1
F16, BF16, F32 = 0, 1, 2

_lib = None


class SamdError(RuntimeError):
    pass


class Params(C.Structure):
    """samd_params_t"""
    _fields_ = [("variant", C.c_int32), ("max_predicts", C.c_int32), ("alpha", C.c_double), ("K", C.c_int32),
                ("len_bias", C.c_int32), ("n_predicts", C.c_int32), ("len_threshold", C.c_int32),
                ("static_null", C.c_int32), ("reserved", C.c_int32)]


class DraftHost(C.Structure):
    """samd_draft_host_t"""
    _fields_ = [("type", C.c_int32), ("n", C.c_int32), ("n_leaves", C.c_int32), ("max_depth", C.c_int32),
                ("index_dyn", C.c_int32), ("match_dyn", C.c_int32), ("index_static", C.c_int32),
                ("match_static", C.c_int32), ("tokens", C.c_int32 * MAX_DRAFT), ("parent", C.c_int32 * MAX_DRAFT),
                ("position", C.c_int32 * MAX_DRAFT), ("mask", C.c_uint64 * MAX_DRAFT), ("mask_hi", C.c_uint64 * MAX_DRAFT),
                ("retrieve", C.c_int32 * (MAX_DRAFT * MAX_DRAFT))]


class E2State(C.Structure):
    """samd_e2_state_t: device pointers of EAGLE-2's tree-logic arrays (include/samd_hip.h)"""
    _NAMES = ("row_lse", "top_logp", "top_idx", "scores", "cs_index", "all_scores", "all_tokens", "parents_list", "mask_rows", "row_src", "ids",
              "rec_top_vals", "rec_top_idx", "rec_best_vals", "rec_best_idx", "rec_final_vals", "rec_final_idx")
    _fields_ = [(n, C.c_void_p) for n in _NAMES]


class Warm(C.Structure):
    """samd_warm_t: the projection that follows a glue launch (L2 warm-up hint, include/samd_hip.h)"""
    _fields_ = [("d_packed_w", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("splits", C.c_int32), ("kb_per_workgroup", C.c_int32),
                ("delay", C.c_int32), ("where", C.c_int32)]


class VerdictHost(C.Structure):
    """samd_verdict_host_t"""
    _fields_ = [("best", C.c_int32), ("accept", C.c_int32), ("next_node", C.c_int32), ("next_token", C.c_int32),
                ("tokens", C.c_int32 * MAX_DRAFT), ("kv_index", C.c_int32 * MAX_DRAFT)]


# every exported symbol of include/samd_hip.h with its signature; tests check this list against the header
_VP, _I32, _I64, _F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_PROTOS = {
    "samd_last_error": (C.c_char_p, []),
    "samd_device_count": (C.c_int, []),
    "samd_host_wait_spin": (C.c_int, [_I32]),
    "samd_device_info": (C.c_int, [_VP]),
    "samd_static_build": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _VP]),
    "samd_static_from_tables": (C.c_int, [_I32, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    "samd_static_save": (C.c_int, [_VP, C.c_char_p]),
    "samd_static_load": (C.c_int, [C.c_char_p, _VP]),
    "samd_static_free": (None, [_VP]),
    "samd_static_upload": (C.c_int, [_VP]),
    "samd_static_info": (C.c_int, [_VP, _VP]),
    "samd_static_from_pickle": (C.c_int, [C.c_char_p, _I32, _VP, _VP]),
    "samd_static_derived_info": (C.c_int, [_VP, _VP]),
    "samd_static_edge_blocks_info": (C.c_int, [_VP, _VP]),
    "samd_static_set_bigram_slots": (C.c_int, [_VP, _I32, _VP]),
    "samd_static_export": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "samd_static_device_image": (C.c_int, [_VP, _VP, _VP]),
    "samd_static_alloc_like": (C.c_int, [_VP, _VP]),
    "samd_static_host_image": (C.c_int, [_VP, _VP, _VP]),
    "samd_static_from_host_image": (C.c_int, [_VP, _VP, _VP]),
    "samd_static_adopt_device": (C.c_int, [_VP, _VP, _VP]),
    "samd_static_walk": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "samd_static_walk_counted": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "samd_static_lookup_batch": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP]),
    "samd_static_walk_streams": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "samd_static_walk_streams_counted": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP]),
    "samd_session_create": (C.c_int, [_I32, _VP]),
    "samd_session_free": (None, [_VP]),
    "samd_session_reset": (C.c_int, [_VP, _VP]),
    "samd_dyn_add_tokens": (C.c_int, [_VP, _VP, _I32, _VP, _VP]),
    "samd_dyn_walk": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP]),
    "samd_session_static_walk": (C.c_int, [_VP, _VP, _VP, _I32, _VP, _I32, _VP, _VP]),
    "samd_session_export": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "samd_session_set_cursors": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _VP]),
    "samd_session_draft": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "samd_session_draft_seq": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_session_draft_tree": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_session_draft_fixed": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_session_set_draft": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP]),
    "samd_session_read_draft": (C.c_int, [_VP, _VP, _VP]),
    "samd_session_set_draft_if_deferred": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP]),
    "samd_session_report_async": (C.c_int, [_VP, _VP, _VP]),
    "samd_session_set_start_token": (C.c_int, [_VP, _VP, _VP]),
    "samd_scripted_argmax": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP]),
    "samd_embed_rows_ssq": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "samd_embed_rows_ssq_rope": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_gemm_qkv_rope_norm": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I64, _I32, _VP]),
    "samd_gemm_qkv_rope_norm_vt": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I64, _I32, _VP]),
    "samd_gemm_pairs_silu_norm": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, _I32, _I32, _I32, _VP, _I32, _VP]),
    "samd_gemm_cs_residual": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP, _VP, _I32, _VP]),
    "samd_gemm_cs_residual_early": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _I32, _VP, _VP, _I32, _VP]),
    "samd_tree_attention_signal": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _F32, _VP, _I64, _VP, _VP]),
    "samd_session_report_target": (C.c_int, [_VP, C.POINTER(_VP)]),
    "samd_report_wait": (C.c_int, [_VP, _I32, _I64]),
    "samd_scripted_logits": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I64, _I32, _VP]),
    "samd_scripted_logits_order1": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I64, _I32, _VP]),
    "samd_session_device_views": (C.c_int, [_VP, _VP]),
    "samd_tree_buffers": (C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "samd_argmax_rows": (C.c_int, [_VP, _I32, _I32, _I64, _I64, _VP, _VP, _VP]),
    "samd_session_accept": (C.c_int, [_VP, _VP, _VP]),
    "samd_session_read_verdict": (C.c_int, [_VP, _VP, _VP]),
    "samd_session_commit": (C.c_int, [_VP, _VP, _VP]),
    "samd_session_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "samd_session_step_given": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "samd_session_candidates": (C.c_int, [_VP, _VP, _VP, _I32, _VP]),
    "samd_session_set_cache_length": (C.c_int, [_VP, _I32, _VP]),
    "samd_session_get_cache_length": (C.c_int, [_VP, _VP, _VP]),
    "samd_kv_compact": (C.c_int, [_VP, _VP, _I32, _I32, _I64, _I32, _I32, _VP]),
    "samd_kv_compact_indices": (C.c_int, [_VP, _I32, _I32, _I64, _I32, _I32, _I32, _VP, _I32, _VP]),
    "samd_tree_attention_workspace": (_I64, [_I32, _I32, _I32]),
    "samd_tree_attention": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _F32,
                                      _VP, _I64, _VP]),
    "samd_tree_attention_vt": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _F32,
                                         _VP, _I64, _VP, _VP]),
    "samd_tree_attention_warm": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _F32,
                                           _VP, _I64, _VP, _VP]),
    "samd_rope_rows": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_attention_block": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _F32, _VP]),
    "samd_tree_attention_rope_workspace": (_I64, [_I32, _I32, _I32]),
    "samd_tree_attention_rope": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I64, _VP, _VP, _VP, _F32, _VP, _I64, _VP]),
    "samd_rope_kv_write_cs": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I64, _I32, _I32, _I64, _VP]),
    "samd_rope_kv_write_cs_vt": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I64, _I32, _I32, _I64, _VP]),
    "samd_rope_kv_write_vt": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _I64, _VP]),
    "samd_kv_compact_vt": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I64, _I32, _I32, _VP]),
    "samd_kv_compact_indices_vt": (C.c_int, [_VP, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _VP, _I32, _VP]),
    "samd_posterior_sampled": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I64, _VP, _I32, _VP, _VP, _VP]),
    "samd_posterior_sampled_nodes": (C.c_int, [_VP, _I32, _VP, _I32, _VP, _I32, _I32, _I64, _VP, _I32, _VP, _VP, _VP]),
    "samd_e2_rowstats_workspace": (C.c_int64, [_I64]),
    "samd_e2_stage_extend": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "samd_e2_rowstats": (C.c_int, [_VP, _I32, _I32, _I64, _I64, _VP, _VP, _I64, _VP]),
    "samd_e2_select": (C.c_int, [_VP, _I32, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _VP]),
    "samd_e2_finish": (C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP, _VP]),
    "samd_sum_partials_bias": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _I32, _I32, _I32, _VP]),
    "samd_embed_rows": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "samd_rmsnorm": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _F32, _I32, _I32, _I64, _VP]),
    "samd_rmsnorm_warm": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _F32, _I32, _I32, _I64, _VP, _VP]),
    "samd_rope_kv_write": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _I64, _VP]),
    "samd_silu_mul": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I64, _VP]),
    "samd_prefill_attention": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I64, C.c_float, _VP]),
    "samd_prefill_attention_vt": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I64, C.c_float, _VP]),
    "samd_gemm_splits": (C.c_int, [_I32, _I32, _I32]),
    "samd_gemm_workspace": (_I64, [_I32, _I32, _I32]),
    "samd_gemm_pack_weights": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "samd_gemm_skinny_silu": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP, _I32, _VP]),
    "samd_gemm_skinny": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _I32, _VP]),
    "samd_gemm_skinny_groups": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _I32, _VP]),
    "samd_gemm_pack_qkv64": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "samd_gemm_pack_groups": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "samd_gemm_pairs_silu": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP, _I32, _VP]),
    "samd_gemm_qkv_rope": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I64, _I32, _VP]),
    "samd_gemm_qkv_rope_vt": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I64, _I32, _VP]),
    "samd_recycle_create": (C.c_int, [_I32, _VP, _VP, _I32, _VP]),
    "samd_recycle_free": (None, [_VP]),
    "samd_recycle_update": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _I64, _I64, _VP]),
    "samd_recycle_draft": (C.c_int, [_VP, _VP, _VP, _VP]),
    "samd_recycle_export": (C.c_int, [_VP, _VP, _VP, _VP]),
}


def lib():
    """Load libsamd_hip.so (built by __graft_entry__.build()).  Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SamdError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)          # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().samd_last_error()
        raise SamdError(f"libsamd_hip error {rc}: {msg.decode() if msg else ''}")


def host_waits_by_spinning(device_index=None):
    """Opt-in, called by entry points (bench.py, the evaluation drivers and CLIs) -- importing the package has no side effect.
    The host waits for the decode step's report once per step; letting that wait spin instead of yield (hipDeviceScheduleSpin)
    takes ~25 us off every step (3.48 -> 3.45 ms, measured A/B in bench.py) at the price of one busy host core per GPU process.
    The flag only takes effect when it is set BEFORE the process creates its HIP context, so call this before any device work.
    It applies to the CURRENT device unless `device_index` is given (then that device also becomes current, as
    hipSetDeviceFlags works on the current device).  torch is imported first on purpose: the process must keep using the HIP
    runtime torch was built with; libsamd_hip.so binds to whichever libamdhip64.so.7 is already loaded.
    SAMD_SPIN_WAIT=0 keeps the runtime's default.  Returns True when the flag was applied; a failure is logged, not hidden."""
    if os.environ.get("SAMD_SPIN_WAIT", "1") == "0":
        return False
    import logging
    log = logging.getLogger("samd_hip")
    try:
        import torch  # noqa: F401
        rc = lib().samd_host_wait_spin(-1 if device_index is None else int(device_index))
    except (ImportError, OSError, AttributeError, SamdError) as e:
        log.warning("host_waits_by_spinning: not applied (%s)", e)
        return False
    if rc != 0:
        msg = lib().samd_last_error()
        log.warning("host_waits_by_spinning: not applied (rc %d: %s)", rc, msg.decode() if msg else "")
    return rc == 0


def require_gpu():
    if lib().samd_device_count() < 1:
        raise SamdError("no MI355X / HIP device visible: the SAM-Decoding hot path has no CPU fallback")


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    """host numpy array or torch tensor or int -> void*"""
    if a is None:
        return None
    if isinstance(a, C.c_void_p):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def torch_dtype_code(dtype):
    import torch
    return {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}[dtype]


class StaticAutomaton:
    """Handle of a corpus suffix automaton (samd_static_t): host image + HBM image."""

    def __init__(self, handle):
        self._h = handle

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def build(cls, batch_tokens, eos_token, kind=KIND_COUNT):
        """StaticSAM.build (samd_sam_only/sam/static_sam.py:31-40; samd/sam/static_sam.py:37-46)."""
        if len(batch_tokens) == 0:
            flat = np.zeros(0, np.int32)
        else:
            flat = np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in batch_tokens])
        off = np.zeros(len(batch_tokens) + 1, np.int64)
        off[1:] = np.cumsum([len(t) for t in batch_tokens])
        return cls.build_flat(flat, off, eos_token, kind)

    @classmethod
    def build_flat(cls, tokens, doc_offsets, eos_token, kind=KIND_COUNT):
        tokens = _np_i32(tokens)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        h = C.c_void_p()
        check(lib().samd_static_build(_ptr(tokens), _ptr(doc_offsets), len(doc_offsets) - 1, int(eos_token), kind, C.byref(h)))
        return cls(h)

    @classmethod
    def from_tables(cls, kind, link, length, aux, deg, edge_tok, edge_dst, text=None):
        arrs = [_np_i32(x) for x in (link, length, aux, deg, edge_tok, edge_dst)]
        t = _np_i32(text) if text is not None else None
        h = C.c_void_p()
        check(lib().samd_static_from_tables(kind, len(arrs[0]), *[_ptr(a) for a in arrs], _ptr(t), 0 if t is None else len(t),
                                            C.byref(h)))
        return cls(h)

    @classmethod
    def from_reference_pickle(cls, path, kind=KIND_COUNT):
        """a pickle written by the reference's dump_sam, read by the native streaming reader (include/samd_hip.h samd_static_from_pickle).
        -> (automaton, {max_predicts, alpha, K, n_predicts, cur_index, cur_length, last, max_length} as pickled); raises SamdError when the
        file is not such a pickle"""
        h = C.c_void_p()
        params = (C.c_double * 8)()
        check(lib().samd_static_from_pickle(os.fsencode(path), int(kind), params, C.byref(h)))
        names = ("max_predicts", "alpha", "K", "n_predicts", "cur_index", "cur_length", "last", "max_length")
        return cls(h), {n: (None if v == -1.0 else v) for n, v in zip(names, params)}

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        check(lib().samd_static_load(os.fsencode(path), C.byref(h)))
        return cls(h)

    def save(self, path):
        check(lib().samd_static_save(self._h, os.fsencode(path)))

    def __del__(self):
        try:
            if self._h:
                lib().samd_static_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- facts ---------------------------------------------------------------------------------
    def info(self):
        out = (C.c_int64 * 8)()
        check(lib().samd_static_info(self._h, out))
        keys = ("n_states", "n_edges", "n_spill", "vocab", "device_bytes", "kind", "n_text", "uploaded")
        return dict(zip(keys, list(out)))

    def derived_info(self):
        """what upload() derived on the device next to the image: bytes of the chain words, of the bigram table (+ root entries, child
        bitmap) and of the top-k counts, and the bigram table's slots (include/samd_hip.h samd_static_derived_info)"""
        out = (C.c_int64 * 6)()
        check(lib().samd_static_derived_info(self._h, out))
        d = dict(zip(("chain_bytes", "bigram_bytes", "topk_count_bytes", "bigram_slots", "edge_table_bytes", "edge_table_slots"), list(out)))
        eb = (C.c_int64 * 4)()                                       # round 6: hot words + edge blocks (they replace the edge table when they fit)
        check(lib().samd_static_edge_blocks_info(self._h, eb))
        d.update(zip(("hot_word_bytes", "edge_block_bytes", "edge_block_slots", "edge_block_states"), list(eb)))
        d["resident_bytes"] = (self.info()["device_bytes"] + d["chain_bytes"] + d["bigram_bytes"] + d["topk_count_bytes"] + d["edge_table_bytes"]
                               + d["hot_word_bytes"] + d["edge_block_bytes"])      # image + derived, per replica
        return d

    def set_bigram_slots(self, slots_per_pair=0):
        """re-size the bigram table (include/samd_hip.h samd_static_set_bigram_slots): 0 = the default (4 per root-child edge); the batched
        walk likes 16"""
        check(lib().samd_static_set_bigram_slots(self._h, int(slots_per_pair), current_stream()))
        return self

    def export(self):
        i = self.info()
        n, ne = i["n_states"], i["n_edges"]
        link, length, aux, deg = (np.empty(n, np.int32) for _ in range(4))
        et, ed = np.empty(ne, np.int32), np.empty(ne, np.int32)
        check(lib().samd_static_export(self._h, *[_ptr(a) for a in (link, length, aux, deg, et, ed)]))
        return dict(link=link, length=length, aux=aux, deg=deg, edge_tok=et, edge_dst=ed)

    def upload(self):
        require_gpu()
        check(lib().samd_static_upload(self._h))
        return self

    def device_image(self):
        ptrs, nbytes = (C.c_void_p * 4)(), (C.c_int64 * 4)()
        check(lib().samd_static_device_image(self._h, ptrs, nbytes))
        return [(ptrs[i], nbytes[i]) for i in range(4)]

    def host_image(self):
        """numpy uint8 views of the four host regions (nodes, root table, spill edges, text)."""
        ptrs, nbytes = (C.c_void_p * 4)(), (C.c_int64 * 4)()
        check(lib().samd_static_host_image(self._h, ptrs, nbytes))
        out = []
        for i in range(4):
            n = int(nbytes[i])
            out.append(np.ctypeslib.as_array((C.c_uint8 * n).from_address(ptrs[i])) if n else np.zeros(0, np.uint8))
        return out

    def info_array(self):
        out = (C.c_int64 * 8)()
        check(lib().samd_static_info(self._h, out))
        return np.array(list(out), dtype=np.int64)

    @classmethod
    def from_host_image(cls, info, regions):
        arr = (C.c_int64 * 8)(*[int(x) for x in info])
        regs = [np.ascontiguousarray(r, dtype=np.uint8) for r in regions]
        ptrs = (C.c_void_p * 4)(*[r.ctypes.data if r.size else None for r in regs])
        h = C.c_void_p()
        check(lib().samd_static_from_host_image(arr, ptrs, C.byref(h)))
        return cls(h)

    @classmethod
    def adopt_device(cls, info, tensors):
        """wrap four device tensors (kept alive by the returned object) as the HBM image."""
        arr = (C.c_int64 * 8)(*[int(x) for x in info])
        ptrs = (C.c_void_p * 4)(*[t.data_ptr() if t.numel() else None for t in tensors])
        h = C.c_void_p()
        check(lib().samd_static_adopt_device(arr, ptrs, C.byref(h)))
        obj = cls(h)
        obj._keep = list(tensors)
        return obj

    # ---- batched walk --------------------------------------------------------------------------
    def walk(self, cursors, tokens, commit=True, trace=None, visited=None):
        """cursors int32 [B,2] (cuda), tokens int32 [T,B] (cuda, time-major)."""
        T, B = tokens.shape
        if visited is not None:
            check(lib().samd_static_walk_counted(self._h, _ptr(cursors), _ptr(tokens), B, T, int(commit), _ptr(visited),
                                                 current_stream()))
        else:
            check(lib().samd_static_walk(self._h, _ptr(cursors), _ptr(tokens), B, T, int(commit), _ptr(trace), current_stream()))


    def lookup_batch(self, cursors, tokens, out, visited=None):
        """StaticSAM.lookup over B cursors (SO/sam/static_sam.py:122-125): cursors int32 [B,2] stay, tokens int32 [T,B] (time-major),
        out int32 [B,2] receives every stream's (index, length)."""
        T, B = tokens.shape
        check(lib().samd_static_lookup_batch(self._h, _ptr(cursors), _ptr(tokens), B, T, _ptr(out), _ptr(visited), current_stream()))

    def walk_streams(self, cursors, tokens, commit=True, trace=None, visited=None):
        """cursors int32 [B,2] (cuda), tokens int32 [B,T] (cuda, stream-major); trace int32 [B,T,2]."""
        B, T = tokens.shape
        if visited is not None:
            check(lib().samd_static_walk_streams_counted(self._h, _ptr(cursors), _ptr(tokens), B, T, int(commit), _ptr(visited), current_stream()))
        else:
            check(lib().samd_static_walk_streams(self._h, _ptr(cursors), _ptr(tokens), B, T, int(commit), _ptr(trace), current_stream()))


class Session:
    """One request stream (samd_session_t): dynamic automaton + cursors + draft + verdict, all in HBM."""

    def __init__(self, max_tokens):
        require_gpu()
        h = C.c_void_p()
        check(lib().samd_session_create(int(max_tokens), C.byref(h)))
        self._h = h
        self.max_tokens = int(max_tokens)
        self._views = None

    def __del__(self):
        try:
            if self._h:
                lib().samd_session_free(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        check(lib().samd_session_reset(self._h, current_stream()))

    def add_tokens(self, d_tokens, n=None, d_n=None):
        check(lib().samd_dyn_add_tokens(self._h, _ptr(d_tokens), d_tokens.numel() if n is None else n, _ptr(d_n), current_stream()))

    def dyn_walk(self, d_tokens, n, commit, d_out=None):
        check(lib().samd_dyn_walk(self._h, _ptr(d_tokens), n, int(commit), _ptr(d_out), current_stream()))

    def static_walk(self, sam, d_tokens, n, commit, d_out=None, d_n=None):
        check(lib().samd_session_static_walk(self._h, sam._h if sam is not None else None, _ptr(d_tokens), n, _ptr(d_n),
                                             int(commit), _ptr(d_out), current_stream()))

    def set_cursors(self, dyn_index, dyn_length, st_index, st_length):
        check(lib().samd_session_set_cursors(self._h, dyn_index, dyn_length, st_index, st_length, current_stream()))

    def draft(self, sam, params, d_start_token):
        check(lib().samd_session_draft(self._h, sam._h if sam is not None else None, C.byref(params), _ptr(d_start_token),
                                       current_stream()))

    def draft_seq(self, params, index, match, start):
        check(lib().samd_session_draft_seq(self._h, C.byref(params), index, match, start, current_stream()))

    def draft_tree(self, sam, params, index, match, start):
        check(lib().samd_session_draft_tree(self._h, sam._h, C.byref(params), index, match, start, current_stream()))

    def draft_fixed(self, sam, params, source, index, start):
        check(lib().samd_session_draft_fixed(self._h, sam._h if sam is not None else None, C.byref(params), source, index, start,
                                             current_stream()))

    def set_draft(self, d_tokens, d_parent, n, type_=1, reverse=False):
        check(lib().samd_session_set_draft(self._h, _ptr(d_tokens), _ptr(d_parent), n, type_ | (256 if reverse else 0),
                                           current_stream()))

    def set_draft_if_deferred(self, d_tokens, d_parent, n, reverse=False):
        check(lib().samd_session_set_draft_if_deferred(self._h, _ptr(d_tokens), _ptr(d_parent), n, int(reverse), current_stream()))

    def set_start_token(self, d_src):
        check(lib().samd_session_set_start_token(self._h, _ptr(d_src), current_stream()))

    def report_target(self):
        """the session's PUSHED-report block (include/samd_hip.h): int32[REPORT_INTS + 1] in host-coherent memory that the step kernel
        writes itself; [REPORT_INTS] is the sequence number.  -> (numpy view, address for samd_report_wait)"""
        out = _VP()
        check(lib().samd_session_report_target(self._h, C.byref(out)))
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(REPORT_INTS + 1,))
        return arr, out

    def report_async(self, h_pinned):
        """enqueue the D2H copy of the per-step report block into a pinned int32[REPORT_INTS] tensor."""
        check(lib().samd_session_report_async(self._h, _ptr(h_pinned), current_stream()))

    def scripted_logits(self, d_argmax, logits, markov_vocab, order=2):
        fn = lib().samd_scripted_logits if order == 2 else lib().samd_scripted_logits_order1
        check(fn(self._h, _ptr(d_argmax), _ptr(logits), torch_dtype_code(logits.dtype), logits.shape[0], logits.stride(0), markov_vocab, current_stream()))

    def scripted_argmax(self, d_target, n_target, vocab, d_out):
        check(lib().samd_scripted_argmax(self._h, _ptr(d_target), n_target, vocab, _ptr(d_out), current_stream()))

    def read_draft(self):
        out = DraftHost()
        check(lib().samd_session_read_draft(self._h, C.byref(out), current_stream()))
        return out

    def accept(self, d_node_argmax):
        check(lib().samd_session_accept(self._h, _ptr(d_node_argmax), current_stream()))

    def read_verdict(self):
        out = VerdictHost()
        check(lib().samd_session_read_verdict(self._h, C.byref(out), current_stream()))
        return out

    def commit(self, sam):
        check(lib().samd_session_commit(self._h, sam._h if sam is not None else None, current_stream()))

    def step(self, sam, params, d_node_argmax):
        check(lib().samd_session_step(self._h, sam._h if sam is not None else None, C.byref(params), _ptr(d_node_argmax),
                                      current_stream()))

    def candidates(self, d_candidates, d_rowmap):
        check(lib().samd_session_candidates(self._h, _ptr(d_candidates), _ptr(d_rowmap), d_rowmap.numel(), current_stream()))

    def step_given(self, sam, params, d_best_accept, d_next_token):
        check(lib().samd_session_step_given(self._h, sam._h if sam is not None else None, C.byref(params), _ptr(d_best_accept), _ptr(d_next_token),
                                            current_stream()))

    def set_cache_length(self, length):
        check(lib().samd_session_set_cache_length(self._h, int(length), current_stream()))

    def get_cache_length(self):
        out = C.c_int32()
        check(lib().samd_session_get_cache_length(self._h, C.byref(out), current_stream()))
        return out.value

    def kv_compact(self, d_tensor_ptrs, n_tensors, n_heads, max_len, head_dim, elem_bytes, n_transposed=0):
        """n_transposed: the last that many tensors of the table are V^T ([head][D][max_len], samd_attention_block's layout)"""
        check(lib().samd_kv_compact_vt(self._h, _ptr(d_tensor_ptrs), n_tensors, n_transposed, n_heads, max_len, head_dim, elem_bytes, current_stream()))

    def device_views(self):
        """raw device pointers of the draft/verdict block (see include/samd_hip.h)."""
        if self._views is None:
            out = (C.c_void_p * 16)()
            check(lib().samd_session_device_views(self._h, out))
            names = ("tokens", "parent", "position", "mask", "retrieve", "dmeta", "verdict", "acc_tokens", "kv_index",
                     "start_token", "cache_length", "history", "counters", "meta")
            self._views = {k: out[i] for i, k in enumerate(names)}
        return self._views

    def export(self, with_edges=True):
        info = (C.c_int64 * 10)()
        check(lib().samd_session_export(self._h, info, None, None, None, None, None, None, None, current_stream()))
        ns, ne, nt = info[0], info[1], info[2]
        link, length, minend, deg = (np.empty(ns, np.int32) for _ in range(4))
        et, ed, text = np.empty(ne, np.int32), np.empty(ne, np.int32), np.empty(nt, np.int32)
        check(lib().samd_session_export(self._h, info, _ptr(link), _ptr(length), _ptr(minend),
                                        _ptr(deg) if with_edges else None, _ptr(et) if with_edges else None,
                                        _ptr(ed) if with_edges else None, _ptr(text), current_stream()))
        keys = ("n_states", "n_edges", "n_text", "last", "max_length", "cur_index", "cur_length", "st_index", "st_length", "error")
        out = dict(zip(keys, list(info)))
        out.update(link=link, length=length, aux=minend, deg=deg, edge_tok=et, edge_dst=ed, text=text)
        return out


class TokenRecycleTable:
    """samd_recycle_t: the [V, 8] successor table of Token Recycle (samd/tree_model/token_recycle/token_recycle.py:18-63)
    plus the static draft tree given as child lists."""

    def __init__(self, vocab, tree):
        require_gpu()
        off = np.zeros(len(tree) + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in tree])
        ch = _np_i32([c for cs in tree for c in cs] or [0])
        h = C.c_void_p()
        check(lib().samd_recycle_create(int(vocab), _ptr(off), _ptr(ch), len(tree), C.byref(h)))
        self._h, self.vocab, self.n_nodes = h, int(vocab), len(tree)

    def __del__(self):
        try:
            if self._h:
                lib().samd_recycle_free(self._h)
                self._h = None
        except Exception:
            pass

    def update(self, d_tokens, d_logits, dtype_code, n, vocab, row_stride, d_n=None):
        check(lib().samd_recycle_update(self._h, _ptr(d_tokens), _ptr(d_logits), dtype_code, n, _ptr(d_n), vocab, row_stride,
                                        current_stream()))

    def draft(self, d_start_token, d_out):
        check(lib().samd_recycle_draft(self._h, _ptr(d_start_token), _ptr(d_out), current_stream()))

    def export(self):
        table = np.empty((self.vocab, 8), np.int32)
        present = np.empty(self.vocab, np.uint8)
        check(lib().samd_recycle_export(self._h, _ptr(table), _ptr(present), current_stream()))
        return table, present

