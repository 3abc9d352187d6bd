"""DynSAM -- the per-request suffix automaton, resident in HBM.

Facade over samd_hip.Session with the method names of samd_sam_only/sam/dyn_sam.py (reference lines cited per
method).  Every call here is a single-wavefront kernel launch followed by a read-back, i.e. the granular/slow
form of the API; SamdModel.generate() drives the same device state through the fused step kernel instead.
"""
from dataclasses import dataclass
from typing import Dict, List

import torch

from ._common import CursorOwner, dev_i32, so_params


class DynSAM(CursorOwner):

    @dataclass
    class SAMState:                       # dyn_sam.py:13-18
        next: Dict[int, int]
        link: int
        length: int
        min_endpos: int

    def __init__(self, max_predicts: int = 40, alpha: float = 4.0, device: str = "cuda", max_tokens: int = None):
        self.max_predicts = max_predicts
        self.alpha = alpha
        self.device = device
        if max_tokens is not None:
            self._own_capacity = int(max_tokens)

    # ---- state views (read-backs; tests and tooling) -----------------------------------------------------
    def _info(self):
        return self._sess().export(with_edges=False)

    @property
    def cur_index(self):
        return int(self._info()["cur_index"])

    @property
    def cur_length(self):
        return int(self._info()["cur_length"])

    @property
    def last(self):
        return int(self._info()["last"])

    @property
    def max_length(self):
        return int(self._info()["max_length"])

    @property
    def input_ids(self) -> List[int]:
        return self._info()["text"].tolist()

    @property
    def states(self):
        e = self._sess().export(with_edges=True)
        out, k = [], 0
        for i in range(int(e["n_states"])):
            d = int(e["deg"][i])
            nxt = dict(zip(e["edge_tok"][k:k + d].tolist(), e["edge_dst"][k:k + d].tolist()))
            k += d
            out.append(DynSAM.SAMState(next=nxt, link=int(e["link"][i]), length=int(e["length"][i]), min_endpos=int(e["aux"][i])))
        return out

    # ---- reference API -------------------------------------------------------------------------------------
    def reset(self):
        """dyn_sam.py:37-43 (also rewinds the static cursor that shares the session, as DraftModel.reset does)."""
        self._sess().reset()

    def add_tokens(self, tokens: List[int]):
        """dyn_sam.py:101-105: per token transfer the cursor, then extend."""
        if len(tokens):
            self._sess().add_tokens(dev_i32(tokens))

    def add_state(self, token: int):
        raise NotImplementedError("add_state without the cursor transfer is not exposed; use add_tokens (dyn_sam.py:101-105)")

    def transfer_tokens(self, tokens: List[int]):
        """dyn_sam.py:107-109"""
        if len(tokens):
            self._sess().dyn_walk(dev_i32(tokens), len(tokens), commit=True)

    def transfer_cur_state(self, token: int):
        self.transfer_tokens([token])

    def lookup(self, token: int):
        """dyn_sam.py:111-114: peek, the cursor does not move."""
        sess = self._sess()
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        sess.dyn_walk(dev_i32([token]), 1, commit=False, d_out=out)
        i, l = out.tolist()
        return i, l

    def transfer_state(self, index: int, length: int, token: int):
        """dyn_sam.py:78-87 from an arbitrary (index, length)."""
        s = self._sess()
        info = s.export(with_edges=False)
        s.set_cursors(index, length, int(info["st_index"]), int(info["st_length"]))
        res = self.lookup(token)
        s.set_cursors(int(info["cur_index"]), int(info["cur_length"]), int(info["st_index"]), int(info["st_length"]))
        return res

    def gen_draft(self, index: int, match_length: int, start_token: int):
        """dyn_sam.py:116-121 -> (seq, {"seq_position_ids": [1, len(seq)]})"""
        s = self._sess()
        s.draft_seq(so_params(self.max_predicts, self.alpha), index, match_length, start_token)
        d = s.read_draft()
        seq = list(d.tokens[:d.n])
        return seq, {"seq_position_ids": torch.arange(0, len(seq), dtype=torch.long, device=self.device).unsqueeze(0)}

    def gen_buffers(self, anc_tree: List[int]):
        """dyn_sam.py:123-155 (duplicate of StaticSAM.gen_buffers)"""
        from .static_sam import gen_buffers
        return gen_buffers(anc_tree, self.device)
