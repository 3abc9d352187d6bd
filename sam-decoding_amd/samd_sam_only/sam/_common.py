"""helpers shared by the DynSAM / StaticSAM facades (both variants)."""
import numpy as np
import torch

import samd_hip


def dev_i32(values, device="cuda"):
    samd_hip.require_gpu()
    return torch.as_tensor(np.asarray(values, dtype=np.int32)).to(device)


def so_params(max_predicts=60, alpha=4.0, K=8, len_bias=0, cap=None):
    """device parameters of the SAM-only variant.  A draft holds at most samd_hip.MAX_DRAFT (128) nodes -- one wavefront builds it two
    nodes per lane, two 64-bit mask words per node, two 64-row verify tiles (include/samd_hip.h SAMD_MAX_DRAFT); larger max_predicts are
    served with 128-node drafts (decoding stays lossless, only the accept lengths of very long matches differ from the reference's).
    `cap`: the largest draft the verifier behind this draft model can run (LlamaRunner.max_draft_rows(): 64 once the row-major matrices are
    released or the attention mode has no two-tile form) -- the draft is clamped HERE, when the session's parameters are made, instead of
    failing inside forward_rows on the first wide draft."""
    p = samd_hip.Params()
    lim = samd_hip.MAX_DRAFT if cap is None else max(1, min(int(cap), samd_hip.MAX_DRAFT))
    p.variant, p.max_predicts, p.alpha, p.K, p.len_bias = 0, min(int(max_predicts), lim), float(alpha), int(K), int(len_bias)
    p.n_predicts, p.len_threshold, p.static_null = 0, 0, 0
    return p


def s_params(n_predicts=40, len_threshold=5, len_bias=5, static_null=False, cap=None):
    p = samd_hip.Params()
    lim = samd_hip.MAX_DRAFT if cap is None else max(1, min(int(cap), samd_hip.MAX_DRAFT))
    p.variant, p.max_predicts, p.alpha, p.K = 1, 0, 0.0, 0
    p.len_bias, p.n_predicts, p.len_threshold, p.static_null = int(len_bias), min(int(n_predicts), lim), int(len_threshold), int(static_null)
    return p


def clamp_to_verifier(draft_model, verifier, asked, what):
    """set draft_model.draft_cap from what `verifier` can run (a runner without row-major matrices, or in attention mode
    'block' / 'split2', serves drafts of at most 64 nodes); warns once per model when the configured draft size is above it"""
    cap = getattr(verifier, "max_draft_rows", None)
    cap = int(cap()) if callable(cap) else samd_hip.MAX_DRAFT
    prev = getattr(draft_model, "draft_cap", None)
    draft_model.draft_cap = cap
    if asked > cap and prev != cap:
        import warnings
        warnings.warn(f"{what} = {asked}: this verifier runs drafts of at most {cap} nodes (row-major projection matrices released, or an "
                      f"attention mode without a two-tile form); drafts are capped at {cap} -- decoding stays lossless, accept lengths of "
                      f"very long matches differ from the reference's", RuntimeWarning, stacklevel=3)
    return cap


def tree_buffers_from_draft(d, device):
    """DraftHost -> the tensors gen_buffers returns (samd_sam_only/sam/static_sam.py:164-180): bool mask [1,1,n,n],
    int64 depths [1,n], int64 retrieve [leaves, max_depth] padded with -1."""
    n = d.n
    rows = np.frombuffer(d.mask, dtype=np.uint64, count=n)
    bits = ((rows[:, None] >> np.arange(min(n, 64), dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    if n > 64:                                           # nodes 64..127: the high mask words
        hi = np.frombuffer(d.mask_hi, dtype=np.uint64, count=n)
        bits = np.concatenate([bits, ((hi[:, None] >> np.arange(n - 64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)], axis=1)
    pos = np.asarray(d.position[:n], dtype=np.int64)
    ret = np.asarray(d.retrieve[:d.n_leaves * d.max_depth], dtype=np.int64).reshape(d.n_leaves, d.max_depth)
    return {
        "tree_attn_mask": torch.from_numpy(bits.reshape(1, 1, n, n)).to(device),
        "tree_position_ids": torch.from_numpy(pos.reshape(1, n)).to(device),
        "tree_retrieve_indices": torch.from_numpy(ret.copy()).to(device),
    }


class CursorOwner:
    """a facade that may own a private Session (stand-alone use) or share the DraftModel's."""
    _session = None
    _own_capacity = 4096

    def _sess(self):
        if self._session is None:
            self._session = samd_hip.Session(self._own_capacity)
        return self._session

    def _bind(self, session):
        self._session = session
