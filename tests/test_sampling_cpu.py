"""eval_posterior's sampling branch and the reference-pickle importer, against fixtures recorded from the imported
reference (tests/golden/make_golden_extra.py).  Both are host-side logic (plain torch / pickle), so they run on CPU."""
import os
import random

import numpy as np
import pytest

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sampling_posterior_matches_reference(golden):
    """same `random` seed -> same accepted prefix, same candidate, same residual distribution (utils.py:142-184).
    Tolerance on sample_p: 1e-6 absolute (fp32 softmax of identical inputs)."""
    from samd_sam_only.utils import SamdGenerationConfig, eval_posterior
    for case in golden("posterior_sampling.json.gz"):
        cfg = SamdGenerationConfig(greedy=False, temperature=case["temperature"], top_p=case["top_p"], top_k=case["top_k"])
        logits = torch.tensor(case["logits"], dtype=torch.float32)
        cand = torch.tensor(case["candidates"])
        random.seed(case["seed"])
        best, acc, sp = eval_posterior(logits, cand, cfg)
        assert (int(best), int(acc)) == (case["best"], case["accept"])
        assert np.allclose(sp.view(-1).numpy(), np.asarray(case["sample_p"], dtype=np.float32), atol=1e-6)


def test_reference_pickle_is_imported(golden):
    """a pickle written by the reference's dump_sam loads through load_sam into the flat image: same states, same
    edges in dict order, same top-k tables as a native build of the same corpus."""
    import samd_sam_only as SO
    from oracle import sam_oracle as O
    meta = golden("ref_static_sam_docs.json.gz")
    sam = SO.load_sam(os.path.join(HERE, "golden", "ref_static_sam.pkl"))
    native = SO.build_sam(meta["docs"], meta["eos"])
    assert len(sam.states) == meta["n_states"] == len(native.states)
    assert [(s.next, s.link, s.length, s.cnt_endpos) for s in sam.states] == [(s.next, s.link, s.length, s.cnt_endpos) for s in native.states]
    assert sam.states_topk_next == native.states_topk_next
    ora = O.StaticSAM.build(meta["docs"], meta["eos"])
    tok, dst, _ = ora.export_topk()
    for i, row in enumerate(sam.states_topk_next):
        assert row == list(zip(tok[i][:len(row)].tolist(), dst[i][:len(row)].tolist()))
