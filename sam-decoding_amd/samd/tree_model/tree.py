"""TreeModel plugin interface (reference: samd/tree_model/tree.py:9-30)."""
from typing import Dict, List, Tuple

import torch


class TreeModel(torch.nn.Module):
    # plugins with fused=True expose device-side hooks so SamdModel's step needs no host round trip
    fused = False

    def __init__(self, samd_config=None, lm_config=None, lm=None, dtype: torch.dtype = None, device: str = None) -> None:
        super().__init__()

    def reset(self):
        raise NotImplementedError

    def update(self, tokens=None, last_hidden_states=None, tree_tokens=None, tree_logits=None, **kwargs):
        raise NotImplementedError

    def gen_draft(self, start_token: int) -> Tuple[List[int], Dict[str, torch.Tensor]]:
        raise NotImplementedError

    def gen_buffers(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError
