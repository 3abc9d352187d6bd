"""Token Recycle (reference: samd/tree_model/token_recycle/token_recycle.py:18-63, utils.py:37-99).

The reference keeps a Python dict token -> top-8 ids and fills a static 61-node tree by walking child lists on the
host.  Here the dict is a dense [V, 8] int32 table in HBM (samd_recycle_t): `update` is a top-8 selection kernel over
the verified rows + an ordered scatter (later rows win), `gen_draft` a level-synchronous gather, both enqueued on the
step's stream.  The table persists across requests (reset() is a no-op, token_recycle.py:33-34).
"""
from typing import Dict, List

import torch

import samd_hip
from ..samd_config import SamdConfig
from .tree import TreeModel

TOPK = samd_hip.TOPK


def parents_of(tree: List[List[int]]) -> List[int]:
    anc = [-1] * len(tree)
    for node, childs in enumerate(tree):
        for c in childs:
            anc[c] = node
    return anc


def gen_buffers(tree: List[List[int]], device) -> Dict[str, torch.Tensor]:
    """token_recycle/utils.py:37-99: float mask [1,1,n,n], depths [1,n], retrieve rows in REVERSED leaf order."""
    n = len(tree)
    par = torch.tensor(parents_of(tree), dtype=torch.int32, device="cuda")
    pos = torch.zeros(n, dtype=torch.int32, device="cuda")
    mask_bool = torch.zeros(n * n, dtype=torch.uint8, device="cuda")
    ret = torch.full((n * n,), -1, dtype=torch.int32, device="cuda")
    shape = torch.zeros(2, dtype=torch.int32, device="cuda")
    samd_hip.check(samd_hip.lib().samd_tree_buffers(samd_hip._ptr(par), n, 1, samd_hip._ptr(pos), None, samd_hip._ptr(mask_bool),
                                                    samd_hip._ptr(ret), samd_hip._ptr(shape), samd_hip.current_stream()))
    nl, md = shape.tolist()
    return {
        "tree_attn_mask": mask_bool.view(1, 1, n, n).to(torch.float32).to(device),
        "tree_position_ids": pos.to(torch.long).view(1, n).to(device),
        "tree_retrieve_indices": ret[:nl * md].to(torch.long).view(nl, md).to(device),
    }


class TokenRecycle(TreeModel):
    fused = True

    def __init__(self, config: SamdConfig, lm, dtype: torch.dtype, device: str) -> None:
        super().__init__()
        self.samd_config = config
        self.dtype = dtype
        self.device = device
        self.tree = config.tree
        if len(self.tree) > samd_hip.MAX_DRAFT:
            raise samd_hip.SamdError(f"draft trees hold at most {samd_hip.MAX_DRAFT} nodes")
        self.parents = parents_of(self.tree)
        vocab = (getattr(getattr(lm, "config", None), "vocab_size", None) or getattr(lm, "vocab", None)
                 or getattr(getattr(lm, "shape", None), "vocab", None))
        if vocab is None:
            raise samd_hip.SamdError("TokenRecycle needs the vocabulary size (lm.config.vocab_size)")
        self.vocab = int(vocab)
        self._table = None

    def table(self) -> samd_hip.TokenRecycleTable:
        if self._table is None:
            self._table = samd_hip.TokenRecycleTable(self.vocab, self.tree)
        return self._table

    @property
    def cache(self) -> Dict[int, List[int]]:
        """the reference's dict view of the table (read-back; tests)."""
        tab, present = self.table().export()
        return {int(t): tab[t].tolist() for t in present.nonzero()[0]}

    def reset(self):
        pass  # the table is shared by all requests of the process

    def update(self, tokens=None, last_hidden_states=None, tree_tokens: torch.Tensor = None, tree_logits: torch.Tensor = None, **kwargs):
        """token_recycle.py:40-48: cache[token] = top-8 of its logits row, later rows win."""
        toks = tree_tokens.reshape(-1).to(device="cuda", dtype=torch.int32).contiguous()
        logits = tree_logits.reshape(toks.numel(), -1)
        if logits.stride(-1) != 1:
            logits = logits.contiguous()
        dt = samd_hip.torch_dtype_code(logits.dtype)
        tab = self.table()
        for r0 in range(0, toks.numel(), 16384):
            r1 = min(toks.numel(), r0 + 16384)
            tab.update(toks[r0:r1], logits[r0:r1], dt, r1 - r0, logits.shape[1], logits.stride(0))

    def gen_draft(self, start_token: int):
        """token_recycle.py:50-60 -> (tree tokens, {})"""
        st = torch.tensor([start_token], dtype=torch.int32, device="cuda")
        out = torch.zeros(len(self.tree), dtype=torch.int32, device="cuda")
        self.table().draft(st, out)
        return out.tolist(), {}

    def gen_buffers(self) -> Dict[str, torch.Tensor]:
        return gen_buffers(self.samd_config.tree, self.device)
