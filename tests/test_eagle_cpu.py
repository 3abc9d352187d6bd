"""EAGLE (v1) static-tree plugin (samd/tree_model/eagle.py) against outputs recorded from the imported reference plugin --
Eagle.update / gen_draft / gen_buffers over EagleModel.topk_genrate (tests/golden/make_golden_eagle.py; tiny random config,
CPU, fp32) -- for the tree the reference ships and for an irregular one that exercises its parent-numbering quirk.
Draft tokens and the static buffers must be identical."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import sam_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    from samd.tree_model.eagle import Eagle, EagleHead, StaticDraftTree
    z = np.load(os.path.join(HERE, "golden", "eagle.npz"))
    cfg = dict(zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()))
    cfg["rms_norm_eps"] = float(z["rms_eps"])
    head = EagleHead(cfg, dtype=torch.float32, device="cpu", bias=True)
    pre = f"{name}:w:"
    head.load_state({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    choices = json.loads(str(z[f"{name}:choices"]))
    head.set_tree(StaticDraftTree(choices))

    class LM:                       # anything exposing lm_head.weight, as the reference's `lm`
        lm_head = torch.nn.Linear(64, 320, bias=False)
    LM.lm_head.weight.data = torch.from_numpy(z[f"{name}:head_weight"])
    return z, choices, Eagle(None, LM, torch.float32, "cpu", head=head)


@pytest.mark.parametrize("name", ["std", "odd"])
def test_static_buffers_match_reference(name):
    z, choices, plug = load(name)
    buf = plug.gen_buffers()
    assert buf["tree_attn_mask"][0, 0].to(torch.uint8).tolist() == z[f"{name}:mask"].tolist()
    assert buf["tree_position_ids"].view(-1).tolist() == z[f"{name}:pos"].tolist()
    assert buf["tree_retrieve_indices"].tolist() == z[f"{name}:retrieve"].tolist()
    assert buf["tree_indices"].tolist() == z[f"{name}:tree_indices"].tolist()
    # the parent array the decode engine installs describes the same tree: the tree-buffer kernel's twin (the oracle)
    # gives the same mask and depths; its retrieve rows are the same root->leaf paths in leaf-index order
    ob = O.gen_buffers(plug.tree.parents)
    assert ob["tree_attn_mask"][0, 0].astype(np.uint8).tolist() == z[f"{name}:mask"].tolist()
    assert ob["tree_position_ids"][0].tolist() == z[f"{name}:pos"].tolist()
    assert sorted(map(tuple, ob["tree_retrieve_indices"].tolist())) == sorted(map(tuple, z[f"{name}:retrieve"].tolist()))


@pytest.mark.parametrize("name", ["std", "odd"])
def test_drafts_match_reference(name):
    z, choices, plug = load(name)
    plug.reset()
    for ci in range(int(z[f"{name}:n_calls"])):
        hs, toks = torch.from_numpy(z[f"{name}:c{ci}:hidden"]), torch.from_numpy(z[f"{name}:c{ci}:tokens_in"])
        if ci == 2:                                  # as recorded: two update() calls before this draft
            plug.update(tokens=toks[:1], last_hidden_states=hs[:1])
            plug.update(tokens=toks[1:], last_hidden_states=hs[1:])
        else:
            plug.update(tokens=toks, last_hidden_states=hs)
        draft, kwargs = plug.gen_draft(int(z[f"{name}:c{ci}:start"]))
        assert kwargs == {}
        assert draft == z[f"{name}:c{ci}:draft"].tolist(), f"draft tokens differ in call {ci}"
        assert plug.accept_tokens is None and plug.accept_hidden_states is None


def test_registered_and_default_tree():
    from samd.tree_model import tree_model_cls
    from samd.samd_config import EAGLE_TREE_CHOICES
    from samd.tree_model.eagle import Eagle, StaticDraftTree
    assert tree_model_cls["eagle"] is Eagle
    z = np.load(os.path.join(HERE, "golden", "eagle.npz"))
    assert sorted(map(tuple, EAGLE_TREE_CHOICES)) == sorted(map(tuple, json.loads(str(z["std:choices"]))))
    t = StaticDraftTree(EAGLE_TREE_CHOICES)
    assert t.n == 26 and t.parents[0] == -1 and all(0 <= t.parents[i] < i for i in range(1, t.n))
