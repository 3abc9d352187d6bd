"""DraftModel of the full variant: SAM sequence drafts when the match is long enough, else the tree-draft plugin.

Same constructor and methods as samd/draft.py:24-79 of the reference.
"""
from collections import namedtuple
from enum import Enum
from typing import Optional

import torch

import samd_hip
from samd_sam_only.sam._common import dev_i32, s_params
from .sam import DynSAM, NullStaticSAM, StaticSAM
from .samd_config import SamdConfig
from .tree_model import TreeModel, tree_model_cls


class CandidateType(str, Enum):
    sequence = "sequence"
    tree = "tree"


Candidates = namedtuple('Candidates', ['type', 'tokens', 'candidate_tokens', 'buffers_kwargs'])

TOPK = samd_hip.TOPK


class DraftModel(torch.nn.Module):

    def __init__(self,
        config: SamdConfig,
        sam_dyn: DynSAM = None,
        sam_static: StaticSAM = None,
        tree_model: TreeModel = None,
        lm=None,
        dtype: torch.dtype = torch.float16,
        device: str = "cuda",
    ) -> None:
        super().__init__()
        tree_cls = tree_model_cls[config.tree_method]
        self.config = config
        self.device = device
        self.sam_dyn = sam_dyn if sam_dyn is not None else DynSAM(config.n_predicts)
        self.sam_static = sam_static if sam_static is not None else NullStaticSAM(config.n_predicts)
        self.tree_model = tree_model if tree_model is not None else tree_cls(config, lm, dtype, device)
        self.sam_dyn.n_predicts = config.n_predicts
        self.sam_static.n_predicts = config.n_predicts
        self.len_bias = config.len_bias
        self.len_threshold = config.len_threshold
        self._start = None

    # ---- device handles used by SamdModel's fused path ----------------------------------------------------------
    def session(self) -> samd_hip.Session:
        s = self.sam_dyn._sess()
        self.sam_static._bind(s)
        return s

    def ensure_capacity(self, max_tokens: int):
        if self.sam_dyn._session is None:
            self.sam_dyn._own_capacity = max(self.sam_dyn._own_capacity, int(max_tokens))
        elif self.sam_dyn._session.max_tokens < max_tokens:
            self.sam_dyn._session = samd_hip.Session(int(max_tokens))
        return self.session()

    def static_automaton(self) -> Optional[samd_hip.StaticAutomaton]:
        return self.sam_static._automaton()

    def params(self) -> samd_hip.Params:
        return s_params(self.config.n_predicts, self.len_threshold, self.len_bias, isinstance(self.sam_static, NullStaticSAM))

    # ---- reference API ----------------------------------------------------------------------------------------------
    def reset(self):
        """draft.py:47-50"""
        self.session().reset()
        self.tree_model.reset()

    def lookup(self, start_token: int):
        """draft.py:52-63"""
        s = self.session()
        self._start = dev_i32([start_token])
        s.draft(self.static_automaton(), self.params(), self._start)
        d = s.read_draft()
        if d.type == 0:
            return (CandidateType.sequence, list(d.tokens[:d.n]), {})
        return (CandidateType.tree,) + tuple(self.tree_model.gen_draft(start_token))

    def update(self,
        tokens: Optional[torch.Tensor] = None,
        last_hidden_states: Optional[torch.Tensor] = None,
        tree_tokens: Optional[torch.Tensor] = None,
        tree_logits: Optional[torch.Tensor] = None,
    ):
        """draft.py:65-79"""
        t = tokens.reshape(-1).to(device="cuda", dtype=torch.int32)
        if t.numel():
            s = self.session()
            s.add_tokens(t)
            s.static_walk(self.static_automaton(), t, t.numel(), commit=True)
        self.tree_model.update(tokens=tokens, last_hidden_states=last_hidden_states, tree_tokens=tree_tokens, tree_logits=tree_logits)
