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
        if not 1 <= self.max_predicts <= MAX_DRAFT:
            raise ValueError(f"max_predicts must be in [1, {MAX_DRAFT}] (one wavefront verifies one draft)")


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
