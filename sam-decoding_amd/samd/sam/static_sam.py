"""StaticSAM / NullStaticSAM of the full variant (reference: samd/sam/static_sam.py:7-137): first end positions +
the corpus text, fixed-length sequence drafts (no back-off)."""
from dataclasses import dataclass
from typing import Dict, List

import samd_hip
from samd_sam_only.sam.static_sam import StaticSAM as _SoStaticSAM
from samd_sam_only.sam._common import s_params


class StaticSAM(_SoStaticSAM):
    KIND = samd_hip.KIND_ENDPOS

    @dataclass
    class SAMState:                       # static_sam.py:10-15
        next: Dict[int, int]
        link: int
        length: int
        min_endpos: int

    def __init__(self, n_predicts: int = 40):
        super().__init__()
        self.n_predicts = n_predicts

    @staticmethod
    def build(batch_tokens: List[List[int]], eos_token: int, verbose: bool = True):
        sam = StaticSAM()
        sam.add_batch_tokens(batch_tokens, eos_token, verbose)
        sam.init_topk_next()
        return sam

    def gen_draft(self, index: int, start_token: int) -> List[int]:
        """static_sam.py:119-125"""
        s = self._sess()
        s.draft_fixed(self._automaton(), s_params(self.n_predicts), 1, index, start_token)
        d = s.read_draft()
        return list(d.tokens[:d.n])


class NullStaticSAM(StaticSAM):
    """static_sam.py:128-137: never matches (the SAM-less configuration of the full variant)."""

    def __init__(self, n_predicts=40):
        super().__init__(n_predicts)

    def _automaton(self):
        return None

    def transfer_tokens(self, tokens):
        pass

    def lookup(self, token):
        return 0, 0

    def gen_draft(self, index, start_token):
        return -1, -1
