"""build_sam / dump_sam / load_sam of the full variant (reference surface: samd/sam/utils.py:10-37): the automaton keeps first
end positions and the corpus text instead of occurrence counts; same flat image format as samd_sam_only."""
from typing import List

from samd_sam_only.sam.utils import load_image_or_pickle
from .static_sam import StaticSAM


def build_sam(batch_tokens: List[List[int]], eos_token: int) -> StaticSAM:
    return StaticSAM.build(batch_tokens, eos_token)


def dump_sam(path: str, sam: StaticSAM) -> None:
    sam.init_topk_next()
    sam._auto.save(path)


def load_sam(path: str) -> StaticSAM:
    return load_image_or_pickle(path, StaticSAM)
