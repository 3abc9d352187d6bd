"""Extra fixtures from the IMPORTED reference (dev container only):
  * posterior_sampling.json.gz -- eval_posterior's sampling branch (samd_sam_only/utils.py:142-184) under a seeded
    `random`, for several temperature / top-p / top-k settings;
  * ref_static_sam.pkl -- a pickle written by the reference's own dump_sam (samd_sam_only/sam/utils.py:20-22) for a tiny
    corpus, to pin load_sam's importer for reference pickles.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py
"""
import gzip
import io
import json
import os
import random
import sys
import types
from contextlib import redirect_stderr, redirect_stdout

import numpy as np
import torch

R = "/root/reference"
sys.path.insert(0, R)
for pkg in ("samd_sam_only", "samd"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{R}/{pkg}"]
    sys.modules[pkg] = m

from samd_sam_only.sam import StaticSAM, dump_sam                                     # noqa: E402
from samd_sam_only.utils import SamdGenerationConfig, eval_posterior                  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(99)
    cases = []
    for ci, (temp, top_p, top_k, C, depth, V) in enumerate([(0.7, 0.0, 0, 4, 5, 40), (1.0, 0.9, 0, 6, 4, 40), (0.5, 0.0, 8, 3, 6, 40),
                                                            (1.3, 0.8, 12, 5, 3, 40), (0.7, 0.0, 0, 1, 7, 40)]):
        cfg = SamdGenerationConfig(greedy=False, temperature=temp, top_p=top_p, top_k=top_k)
        for rep in range(4):
            logits = torch.tensor(rng.normal(size=(C, depth, V)).astype(np.float32) * 2.5)
            # candidates share prefixes like retrieve-gathered tree paths do; -1 marks padding
            base = rng.integers(0, V, depth)
            cand = np.tile(base, (C, 1))
            for c in range(1, C):
                cut = int(rng.integers(1, depth))
                cand[c, cut:] = rng.integers(0, V, depth - cut)
                if rng.random() < 0.3:
                    cand[c, -1] = -1
            # make the likely tokens appear among the candidates so that some get accepted
            for c in range(C):
                for d in range(1, depth):
                    if cand[c, d] >= 0 and rng.random() < 0.6:
                        cand[c, d] = int(logits[c, d - 1].argmax())
            cand_t = torch.tensor(cand)
            seed = 1000 * ci + rep
            random.seed(seed)
            best, acc, sp = eval_posterior(logits, cand_t, cfg)
            cases.append({"temperature": temp, "top_p": top_p, "top_k": top_k, "seed": seed, "logits": logits.numpy().round(6).tolist(),
                          "candidates": cand.tolist(), "best": int(best), "accept": int(acc), "sample_p": sp.view(-1).numpy().tolist()})
    with gzip.open(os.path.join(HERE, "posterior_sampling.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("posterior_sampling cases:", len(cases), "accept lengths:", [c["accept"] for c in cases])

    docs = [rng.integers(3, 30, 40).tolist() for _ in range(4)] + [[i] for i in range(30)]
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        sam = StaticSAM.build(docs, 2, False)
    sam.cur_index, sam.cur_length = 5, 2                  # stale cursor: load_sam copies it, generate() resets it
    dump_sam(os.path.join(HERE, "ref_static_sam.pkl"), sam)
    with gzip.open(os.path.join(HERE, "ref_static_sam_docs.json.gz"), "wt") as f:
        json.dump({"docs": docs, "eos": 2, "n_states": len(sam.states)}, f)
    print("ref_static_sam.pkl", os.path.getsize(os.path.join(HERE, "ref_static_sam.pkl")), "bytes,", len(sam.states), "states")


if __name__ == "__main__":
    main()
