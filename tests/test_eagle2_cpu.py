"""EAGLE-2 tree expansion (samd/tree_model/eagle2.py: Eagle2Head.topk_generate) against outputs recorded from the
imported reference Eagle2Model.topk_genrate (tests/golden/make_golden_eagle2.py; tiny random config, CPU, fp32).
Draft tokens must be identical; mask / positions / retrieve rows built from OUR parent array by the oracle's gen_buffers
must equal the tensors the reference built with its host loops (eagle2_model.py:915-946)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import sam_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def load_head():
    from samd.tree_model.eagle2 import Eagle2Head
    z = np.load(os.path.join(HERE, "golden", "eagle2.npz"))
    cfg = dict(zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()))
    cfg["rms_norm_eps"] = float(z["rms_eps"])
    head = Eagle2Head(cfg, dtype=torch.float32, device="cpu", bias=True)
    head.load_state({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    return z, head


def test_topk_generate_matches_reference():
    z, head = load_head()
    lm_head = torch.from_numpy(z["head_weight"])
    head.reset()
    for ci in range(int(z["n_calls"])):
        toks, parents = head.topk_generate(torch.from_numpy(z[f"c{ci}:hidden"]), torch.from_numpy(z[f"c{ci}:ids"]), lm_head)
        assert toks.tolist() == z[f"c{ci}:tokens"].tolist(), f"draft tokens differ in call {ci}"
        par = parents.tolist()
        assert par[0] == -1 and all(0 <= par[i] < i for i in range(1, len(par)))
        buf = O.gen_buffers(par)
        assert buf["tree_attn_mask"][0, 0].astype(np.uint8).tolist() == z[f"c{ci}:mask"].tolist()
        assert buf["tree_position_ids"][0].tolist() == z[f"c{ci}:pos"].tolist()
        assert buf["tree_retrieve_indices"].tolist() == z[f"c{ci}:retrieve"].tolist()


def test_plugin_accumulates_and_consumes_state():
    """Eagle2.update / gen_draft_device bookkeeping (eagle2.py:37-63) without a GPU."""
    from samd.tree_model.eagle2 import Eagle2
    z, head = load_head()

    class LM:                       # anything exposing lm_head.weight, as the reference's `lm`
        lm_head = torch.nn.Linear(64, 320, bias=False)
    LM.lm_head.weight.data = torch.from_numpy(z["head_weight"])
    plug = Eagle2(None, LM, torch.float32, "cpu", head=head)
    plug.reset()
    ids0, hs0 = torch.from_numpy(z["c0:ids"]), torch.from_numpy(z["c0:hidden"])
    # the prompt arrives in two pieces (prefill chunks); the start token closes the sequence
    plug.update(tokens=ids0[:5], last_hidden_states=hs0[:5])
    plug.update(tokens=ids0[5:13], last_hidden_states=hs0[5:13])
    toks, parents = plug.gen_draft_device(ids0[13:14])
    assert toks.tolist() == z["c0:tokens"].tolist()
    assert plug.accept_tokens is None and plug.accept_hidden_states is None
    ids1, hs1 = torch.from_numpy(z["c1:ids"]), torch.from_numpy(z["c1:hidden"])
    plug.update(tokens=ids1[:1], last_hidden_states=hs1)
    toks, _ = plug.gen_draft_device(ids1[1:2])
    assert toks.tolist() == z["c1:tokens"].tolist()
    assert plug.gen_buffers() == {"tree_attn_mask": None, "tree_position_ids": None, "tree_retrieve_indices": None}
