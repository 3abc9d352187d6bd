"""Generates tests/golden/ref_timings.json: the IMPORTED reference's Python SAM path (samd_sam_only: StaticSAM / DynSAM /
DraftModel, /root/reference/samd_sam_only/sam/*.py, draft.py) timed in THIS dev container (CPython, 8 host cores, no GPU,
device="cpu") on bench.py's own synthetic generators -- the "reference CPython, dev container" column that bench.py shows
next to its C port (`cpu_baseline.reference_cpython`).  The reference's Python cannot travel to the GPU box, so these
numbers are a fixture, not a same-box measurement; they are labelled as such wherever they are printed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ref_timings.py
"""
import json
import os
import platform
import statistics
import sys
import time
import types

import numpy as np
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, R)
sys.path.insert(0, ROOT)
for pkg in ("samd_sam_only", "samd"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{R}/{pkg}"]
    sys.modules[pkg] = m

from samd_sam_only.samd_config import SamdConfig                     # noqa: E402
from samd_sam_only.sam import DynSAM, StaticSAM                       # noqa: E402
from samd_sam_only.draft import CandidateType, DraftModel            # noqa: E402
import bench                                                          # noqa: E402  (synthetic corpus / request generators only)

# bench.py's own corpus size by default (2^22 tokens: ~2 minutes of CPython and ~10 GB of Python objects for the build; round 2 used
# 2^17, which made this column a different workload from the port's and the GPU's).  REF_CORPUS_TOKENS overrides.
CORPUS_TOKENS = int(os.environ.get("REF_CORPUS_TOKENS", 1 << 22))
EOS = 2


def main():
    torch.set_num_threads(1)
    flat, off, docs = bench.synth_corpus(CORPUS_TOKENS)
    batch = [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    n_tok = int(off[-1])
    t = time.perf_counter()
    sam = StaticSAM.build(batch, EOS, verbose=False)
    build_s = time.perf_counter() - t
    cfg = SamdConfig(max_predicts=60, alpha=4.0, K=8, len_bias=0)
    dm = DraftModel(cfg, sam_static=sam, device="cpu")
    rng = np.random.default_rng(1000)
    t_add = t_walk = t_lookup = t_update = 0.0
    n_add = n_lookup = n_update_tok = steps = tokens = 0
    tree_ms, seq_ms = [], []
    t_wall = time.perf_counter()
    while time.perf_counter() - t_wall < 25.0:
        prompt, target = bench.synth_request(rng, docs)
        dm.reset()
        t = time.perf_counter(); dm.sam_dyn.add_tokens(prompt); t_add += time.perf_counter() - t; n_add += len(prompt)
        t = time.perf_counter(); dm.sam_static.transfer_tokens(prompt); t_walk += time.perf_counter() - t
        pos = len(prompt)
        while pos < len(prompt) + 512 and pos + 70 < len(target):
            t = time.perf_counter()
            ty, tok, buf = dm.lookup(target[pos])
            dt = time.perf_counter() - t
            t_lookup += dt; n_lookup += 1
            (tree_ms if ty == CandidateType.tree else seq_ms).append(dt * 1e3)
            # scripted greedy verdict (the LM stand-in, not timed): longest root->node path that follows the continuation
            if ty == CandidateType.tree:
                ret = buf["tree_retrieve_indices"].tolist()
                best = 1
                for row in ret:
                    k = 1
                    while k < len(row) and row[k] >= 0 and tok[row[k]] == target[pos + k]:
                        k += 1
                    best = max(best, k)
            else:
                best = 1
                while best < len(tok) and tok[best] == target[pos + best]:
                    best += 1
            acc = target[pos:pos + best]
            t = time.perf_counter(); dm.update(torch.tensor(acc)); t_update += time.perf_counter() - t
            n_update_tok += len(acc); pos += len(acc); steps += 1; tokens += len(acc)
    # isolated pieces
    probe = [int(x) for x in rng.integers(3, 32000, 2000)]
    t = time.perf_counter()
    for x in probe:
        dm.sam_dyn.lookup(x); dm.sam_static.lookup(x)
    pair_us = (time.perf_counter() - t) / len(probe) * 1e6
    # StaticSAM.gen_draft + gen_buffers at the full draft size n = 60 (match length 15 -> 1 + int(15 * 4) capped at 60)
    n60 = []
    for _ in range(200):
        d, s0 = int(rng.integers(0, docs.shape[0])), int(rng.integers(0, docs.shape[1] - 24))
        sam.reset()
        sam.transfer_tokens(docs[d, s0:s0 + 20].tolist())
        idx, ln = sam.lookup(int(docs[d, s0 + 20]))
        t = time.perf_counter()
        tree, buf = sam.gen_draft(idx, 15, int(docs[d, s0 + 20]))
        if len(tree) == 60:
            n60.append((time.perf_counter() - t) * 1e3)
    sam.reset()
    out = {
        "label": "reference CPython, dev container (NOT the GPU box)", "cores": os.cpu_count(), "threads_used": 1,
        "python": platform.python_version(), "torch": torch.__version__, "cpu": platform.processor() or platform.machine(),
        "inputs": f"bench.synth_corpus({CORPUS_TOKENS}) + 32000 one-token documents; bench.synth_request streams, seed 1000; "
                  "max_predicts 60, alpha 4, len_bias 0, K 8; tensors on device='cpu'",
        "static_states": len(sam.states),
        "static_build_us_per_token": round(build_s / n_tok * 1e6, 2),
        "dyn_add_tokens_us_per_token": round(t_add / n_add * 1e6, 2),
        "static_transfer_tokens_us_per_token": round(t_walk / n_add * 1e6, 2),
        "lookup_dyn_plus_static_us": round(pair_us, 2),
        "draftmodel_lookup_tree_ms_median": round(statistics.median(tree_ms), 3) if tree_ms else None,
        "draftmodel_lookup_tree_ms_max": round(max(tree_ms), 3) if tree_ms else None,
        "draftmodel_lookup_tree_ms_p90": round(float(np.percentile(tree_ms, 90)), 3) if tree_ms else None,
        "static_gen_draft_plus_buffers_n60_ms_median": round(statistics.median(n60), 3) if n60 else None,
        "static_gen_draft_plus_buffers_n60_ms_max": round(max(n60), 3) if n60 else None, "n60_samples": len(n60),
        "draftmodel_lookup_seq_ms_median": round(statistics.median(seq_ms), 3) if seq_ms else None,
        "draftmodel_update_us_per_token": round(t_update / max(n_update_tok, 1) * 1e6, 2),
        "loop_steps": steps, "loop_tokens": tokens, "loop_mean_accept": round(tokens / max(steps, 1), 3),
        "loop_us_per_step": round((t_lookup + t_update) / max(steps, 1) * 1e6, 1),
        "loop_tokens_per_s": round(tokens / max(t_lookup + t_update, 1e-9), 1),
        "tree_steps": len(tree_ms), "seq_steps": len(seq_ms),
    }
    with open(os.path.join(HERE, "ref_timings.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
