"""DynSAM of the full variant (reference: samd/sam/dyn_sam.py:17-113): same automaton as samd_sam_only's, but drafts are
fixed-length (`n_predicts`, zero padded) after backing off along suffix links while the continuation is too short."""
from typing import List

import torch

import samd_hip
from samd_sam_only.sam import DynSAM as _SoDynSAM
from samd_sam_only.sam._common import s_params


class DynSAM(_SoDynSAM):

    def __init__(self, n_predicts: int = 40, device: str = "cuda", max_tokens: int = None):
        super().__init__(device=device, max_tokens=max_tokens)
        self.n_predicts = n_predicts

    def to_anc(self, index: int):
        """dyn_sam.py:99-105: computed inside the fixed-length draft kernel, which records the backed-off state in the
        draft's meta block (word 9)."""
        s = self._sess()
        s.draft_fixed(None, s_params(self.n_predicts), 0, index, 0)
        rep = torch.zeros(samd_hip.REPORT_INTS, dtype=torch.int32).pin_memory()
        s.report_async(rep)
        torch.cuda.current_stream().synchronize()
        return int(rep[samd_hip.REP_DMETA + 9])

    def gen_draft(self, index: int, start_token: int) -> List[int]:
        """dyn_sam.py:107-113"""
        s = self._sess()
        s.draft_fixed(None, s_params(self.n_predicts), 0, index, start_token)
        d = s.read_draft()
        return list(d.tokens[:d.n])
