"""SamdConfig and the forward-state carriers of the SAM-only variant.

Mirrors samd_sam_only/samd_config.py:9-37 of the reference (same dataclass fields and defaults, same enum values) so
that evaluation/inference_sam_only.py constructs it unchanged.
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import Literal, Optional

import torch


@dataclass
class SamdConfig:
    max_predicts: int = field(default=60)
    alpha: float = field(default=4.0)
    K: int = field(default=8)
    len_bias: int = field(default=5)
    cache_type: Literal["dynamic", "static"] = field(default="static")

    def __post_init__(self):
        from samd_hip import MAX_DRAFT
        if self.max_predicts < 1:
            raise ValueError("max_predicts must be >= 1")
        if self.max_predicts > MAX_DRAFT:
            # the reference takes any value (samd_sam_only/sam/static_sam.py:183); here one wavefront builds and verifies one draft, so
            # drafts are capped at MAX_DRAFT nodes.  Decoding is lossless either way -- the same tokens come out -- only the accept
            # lengths of matches longer than (MAX_DRAFT - 1) / alpha tokens differ.  INTEGRATION.md section A states the limit.
            import warnings
            warnings.warn(f"max_predicts = {self.max_predicts}: drafts are capped at {MAX_DRAFT} nodes on this implementation "
                          "(output tokens are unaffected)", RuntimeWarning, stacklevel=2)


class ForwardType(str, Enum):
    prefill = "prefill"
    seq_decode = "seq_decode"
    tree_decode = "tree_decode"


class ForwardState:
    """which kind of forward is in flight (reference: samd_config.py:25-28); informational here, the mask rides
    with the draft block in HBM."""

    def __init__(self, forward_type: Optional[ForwardType]) -> None:
        self.forward_type = forward_type


class MaskState:
    """holder of the current tree mask tensor (reference: samd_config.py:31-37)."""

    def __init__(self, mask: Optional[torch.Tensor]) -> None:
        self.mask = mask

    def set_state(self, mask: Optional[torch.Tensor]) -> None:
        self.mask = mask
