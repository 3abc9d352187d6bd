"""DraftModel -- arbitrates between the dynamic and the static suffix automaton.

Same constructor and methods as samd_sam_only/draft.py:22-67.  The two automata share one samd_hip.Session (the
DynSAM's arena also carries the static cursor, the draft block and the verdict), which is what lets
SamdModel.generate() run lookup -> draft -> buffers -> accept -> update as one kernel per step.
"""
from collections import namedtuple
from enum import Enum
from typing import Optional

import torch

import samd_hip
from .sam import DynSAM, StaticSAM
from .sam._common import dev_i32, so_params, tree_buffers_from_draft
from .samd_config import SamdConfig


class CandidateType(str, Enum):
    sequence = "sequence"
    tree = "tree"


Candidates = namedtuple('Candidates', ['type', 'tokens', 'candidate_tokens', 'buffers_kwargs'])

TOPK = samd_hip.TOPK


class SessionPlumbing:
    """what both variants' DraftModel need to hand to SamdModel's fused path: the shared device session (the DynSAM's
    arena, which also carries the static cursor, the draft block and the verdict) and the uploaded static automaton."""

    def session(self) -> samd_hip.Session:
        s = self.sam_dyn._sess()
        if self.sam_static is not None:
            self.sam_static._bind(s)
        return s

    def ensure_capacity(self, max_tokens: int):
        """the dynamic automaton's arena is bounded (prompt + generated tokens <= max_cache_len)."""
        if self.sam_dyn._session is None:
            self.sam_dyn._own_capacity = max(self.sam_dyn._own_capacity, int(max_tokens))
        elif self.sam_dyn._session.max_tokens < max_tokens:
            self.sam_dyn._session = samd_hip.Session(int(max_tokens))
        return self.session()

    def static_automaton(self) -> Optional[samd_hip.StaticAutomaton]:
        return None if self.sam_static is None else self.sam_static._automaton()

    def _extend(self, tokens: torch.Tensor):
        """DynSAM.add_tokens + StaticSAM.transfer_tokens of the same tokens (draft.py:62-67)."""
        t = tokens.reshape(-1).to(device="cuda", dtype=torch.int32)
        if t.numel():
            s = self.session()
            s.add_tokens(t)
            s.static_walk(self.static_automaton(), t, t.numel(), commit=True)


class DraftModel(SessionPlumbing, torch.nn.Module):

    def __init__(self,
        config: SamdConfig,
        sam_dyn: DynSAM = None,
        sam_static: StaticSAM = None,
        lm=None,
        dtype: torch.dtype = torch.float16,
        device: str = "cuda",
        max_tokens: int = None,
    ) -> None:
        super().__init__()
        self.config = config
        self.device = device
        self.sam_dyn = sam_dyn if sam_dyn is not None else DynSAM(config.max_predicts, config.alpha, device, max_tokens=max_tokens)
        self.sam_static = sam_static          # None = empty automaton: never matches (draft.py:35 builds a root-only one)
        self.sam_dyn.max_predicts = config.max_predicts
        self.sam_dyn.alpha = config.alpha
        if self.sam_static is not None:
            self.sam_static.max_predicts = config.max_predicts
            self.sam_static.alpha = config.alpha
            self.sam_static.K = config.K
            self.sam_static.device = device
        self.len_bias = config.len_bias
        self._start = None

    def params(self) -> samd_hip.Params:
        c = self.config
        return so_params(c.max_predicts, c.alpha, c.K, self.len_bias, cap=getattr(self, "draft_cap", None))

    # ---- reference API ----------------------------------------------------------------------------------------------
    def reset(self):
        """draft.py:45-47"""
        self.session().reset()

    def lookup(self, start_token: int):
        """draft.py:50-59 -> (CandidateType, tokens, buffers_kwargs); one kernel does both lookups, the rule, the
        draft and its buffers."""
        s = self.session()
        self._start = dev_i32([start_token])
        s.draft(self.static_automaton(), self.params(), self._start)
        d = s.read_draft()
        tokens = list(d.tokens[:d.n])
        if d.type == 0:
            return (CandidateType.sequence, tokens,
                    {"seq_position_ids": torch.arange(0, d.n, dtype=torch.long, device=self.device).unsqueeze(0)})
        return (CandidateType.tree, tokens, tree_buffers_from_draft(d, self.device))

    def update(self, tokens: Optional[torch.Tensor] = None):
        """draft.py:62-67: dyn add_tokens + static transfer_tokens of the accepted tokens."""
        self._extend(tokens)

    def prefill_update(self, tokens: Optional[torch.Tensor] = None):
        self.update(tokens)

