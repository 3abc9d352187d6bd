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
        sam = load_reference_pickle(path, cls)
        sam.init_topk_next()
    print("loading ended in {} seconds.".format(time.perf_counter() - start))
    return sam


def load_reference_pickle(path: str, cls):
    """a pickle written by the reference's dump_sam (SO/sam/utils.py:20-22).  First the native streaming reader
    (samd_static_from_pickle: the states go straight into flat tables, peak memory ~ the node image, the native reader executes nothing of
    the stream -- the published 20-35 M-state automata would otherwise become ~10^8 live Python objects); a pickle outside its opcode subset
    falls back to pickle.load + cls.__setstate__ (small files, odd protocols), with a warning that says why.  NOTE: that fallback is
    CPython's unpickler and DOES execute the stream, exactly as the reference's load_sam does (SO/sam/utils.py:24-39) -- only load files
    you trust, or set SAMD_PICKLE_NO_FALLBACK=1 to turn a decline into an error instead."""
    try:
        auto, pickled = samd_hip.StaticAutomaton.from_reference_pickle(path, cls.KIND)
    except samd_hip.SamdError as e:
        import os
        import warnings
        if os.environ.get("SAMD_PICKLE_NO_FALLBACK", "0") == "1":
            raise
        warnings.warn(f"native pickle reader declined {path} ({e}); falling back to pickle.load", RuntimeWarning)
        with open(path, "rb") as f:
            sam = pickle.load(f)
        assert type(sam) is cls
        return sam
    sam = cls._from_automaton(auto)
    for key in ("max_predicts", "alpha", "K", "n_predicts"):            # utils.py:29-33 copies the pickled attributes a fresh object also has
        v = pickled.get(key)
        if v is not None and hasattr(sam, key):
            setattr(sam, key, type(getattr(sam, key))(v))
    return sam


def load_sam(path: str):
    return load_image_or_pickle(path, StaticSAM)
