"""DraftModel of the full variant (reference surface: samd/draft.py:24-79).

Rule (draft.py:52-63): when the longer of the two matches -- the static one after `len_bias` -- reaches `len_threshold`,
draft a fixed-length sequence from the automaton that matched longer (dynamic wins ties); otherwise hand over to the
tree-draft plugin.  The decision and the sequence draft are one kernel (sam_device.h: do_draft, variant 1), which
reports "deferred" as draft type 2."""
from collections import namedtuple
from enum import Enum
from typing import Optional

import torch

import samd_hip
from samd_sam_only.draft import SessionPlumbing
from samd_sam_only.sam._common import dev_i32, s_params
from .sam import DynSAM, NullStaticSAM, StaticSAM
from .samd_config import SamdConfig
from .tree_model import TreeModel, tree_model_cls

TOPK = samd_hip.TOPK
Candidates = namedtuple('Candidates', ['type', 'tokens', 'candidate_tokens', 'buffers_kwargs'])


class CandidateType(str, Enum):
    sequence = "sequence"
    tree = "tree"


class DraftModel(SessionPlumbing, torch.nn.Module):

    def __init__(self, config: SamdConfig, sam_dyn: DynSAM = None, sam_static: StaticSAM = None, tree_model: TreeModel = None, lm=None,
                 dtype: torch.dtype = torch.float16, device: str = "cuda") -> None:
        super().__init__()
        self.config, self.device = config, device
        n = config.n_predicts
        self.sam_dyn = DynSAM(n) if sam_dyn is None else sam_dyn
        self.sam_static = NullStaticSAM(n) if sam_static is None else sam_static
        self.tree_model = tree_model_cls[config.tree_method](config, lm, dtype, device) if tree_model is None else tree_model
        self.sam_dyn.n_predicts = self.sam_static.n_predicts = n
        self.len_bias, self.len_threshold = config.len_bias, config.len_threshold
        self._start = None

    def params(self) -> samd_hip.Params:
        return s_params(self.config.n_predicts, self.len_threshold, self.len_bias, isinstance(self.sam_static, NullStaticSAM),
                        cap=getattr(self, "draft_cap", None))

    def reset(self):
        self.session().reset()                       # dynamic automaton wiped, static cursor to the root
        self.tree_model.reset()

    def lookup(self, start_token: int):
        s = self.session()
        self._start = dev_i32([start_token])
        s.draft(self.static_automaton(), self.params(), self._start)
        d = s.read_draft()
        if d.type == 2:                              # deferred: the plugin drafts (draft.py:63)
            return (CandidateType.tree,) + tuple(self.tree_model.gen_draft(start_token))
        return (CandidateType.sequence, list(d.tokens[:d.n]), {})

    def update(self, tokens: Optional[torch.Tensor] = None, last_hidden_states: Optional[torch.Tensor] = None,
               tree_tokens: Optional[torch.Tensor] = None, tree_logits: Optional[torch.Tensor] = None):
        self._extend(tokens)
        self.tree_model.update(tokens=tokens, last_hidden_states=last_hidden_states, tree_tokens=tree_tokens, tree_logits=tree_logits)
