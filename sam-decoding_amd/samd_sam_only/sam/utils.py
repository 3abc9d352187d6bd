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


def load_sam(path: str):
    print("load sam...")
    start = time.perf_counter()
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"SAMDHIP1":
        sam = StaticSAM._from_automaton(samd_hip.StaticAutomaton.load(path))
    else:
        with open(path, "rb") as f:
            sam = pickle.load(f)
        assert type(sam) is StaticSAM
        sam.init_topk_next()
    print("loading ended in {} seconds.".format(time.perf_counter() - start))
    return sam
