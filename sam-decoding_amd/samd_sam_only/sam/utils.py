"""build_sam / dump_sam / load_sam (reference: samd_sam_only/sam/utils.py:10-39).

The automaton is persisted as the flat SAMDHIP1 image the kernels walk (header + 64-byte nodes + root table + spill
edges), which loads with a handful of freads instead of unpickling tens of millions of Python objects.  load_sam
also accepts a pickle written by the reference's dump_sam: its classes resolve to ours (same module path and class
names) and StaticSAM.__setstate__ converts the object graph to tables.
"""
import pickle
import time
from typing import List

import samd_hip
from .static_sam import StaticSAM


def build_sam(batch_tokens: List[List[int]], eos_token: int):
    return StaticSAM.build(batch_tokens, eos_token)


def dump_sam(path: str, sam: StaticSAM):
    sam.init_topk_next()
    sam._auto.save(path)


def load_image_or_pickle(path: str, cls):
    """a SAMDHIP1 image, or a pickle of the reference's object graph that unpickles into `cls` (its __setstate__ converts)."""
    print("load sam...")
    start = time.perf_counter()
    with open(path, "rb") as f:
        is_image = f.read(8) == b"SAMDHIP1"
    if is_image:
        sam = cls._from_automaton(samd_hip.StaticAutomaton.load(path))
    else:
        with open(path, "rb") as f:
            sam = pickle.load(f)
        assert type(sam) is cls
        sam.init_topk_next()
    print("loading ended in {} seconds.".format(time.perf_counter() - start))
    return sam


def load_sam(path: str):
    return load_image_or_pickle(path, StaticSAM)
