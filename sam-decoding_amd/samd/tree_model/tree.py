"""The tree-draft plugin contract (reference surface: samd/tree_model/tree.py:9-30).

A plugin is consulted when no suffix-automaton match is long enough (samd/draft.py:63).  It sees every verified token
with its logits and, if it asks for them, the base model's last hidden states (`update`), and answers `gen_draft` with
the tokens of a draft tree plus the tree's buffers (empty dict = use the static buffers from `gen_buffers`).

`fused` marks plugins whose update/draft are kernels that run inside the decode step's hipGraph (Token Recycle); the
others (EAGLE-2) are called between graph replays by samd_hip.engine.TreeModelEngine."""
import abc
from typing import Dict, List, Tuple

import torch


class TreeModel(torch.nn.Module, metaclass=abc.ABCMeta):
    fused = False

    def __init__(self, samd_config=None, lm_config=None, lm=None, dtype: torch.dtype = None, device: str = None) -> None:
        super().__init__()

    @abc.abstractmethod
    def reset(self):
        """called by DraftModel.reset() at the start of every request."""

    @abc.abstractmethod
    def update(self, tokens=None, last_hidden_states=None, tree_tokens=None, tree_logits=None, **kwargs):
        """accepted tokens (+ hidden states) and all verified tokens (+ logits) of one step, or of the prompt."""

    @abc.abstractmethod
    def gen_draft(self, start_token: int) -> Tuple[List[int], Dict[str, torch.Tensor]]:
        """-> (tree tokens with the start token first, buffers_kwargs)."""

    @abc.abstractmethod
    def gen_buffers(self) -> Dict[str, torch.Tensor]:
        """static base buffers: tree_attn_mask, tree_position_ids, tree_retrieve_indices (None when the tree is dynamic)."""
