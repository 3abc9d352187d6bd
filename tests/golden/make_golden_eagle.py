"""Generates tests/golden/eagle.npz by running the IMPORTED reference EAGLE (v1) plugin -- Eagle.update / gen_draft /
gen_buffers (samd/tree_model/eagle/eagle.py:13-75) over EagleModel.topk_genrate (eagle_model.py:783-845) with the static
tree of samd/config/eagle.json -- on a tiny random configuration, CPU, fp32.  Dev-container only (needs /root/reference);
the fixture holds weights, the tree choices, inputs and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eagle.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

R = "/root/reference"
sys.path.insert(0, R)
for pkg in ("samd_sam_only", "samd"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{R}/{pkg}"]
    sys.modules[pkg] = m

from samd.tree_model.eagle.eagle_config import EagleConfig            # noqa: E402
from samd.tree_model.eagle.eagle_model import EagleModel              # noqa: E402
from samd.tree_model.eagle.eagle import Eagle                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(4321)
    cfg_kw = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-6, pad_token_id=0)
    cfg = EagleConfig(**cfg_kw)
    cfg.rope_scaling = None                      # transformers 5.x injects a default dict the 2023 code cannot parse
    tree = json.load(open(f"{R}/samd/config/eagle.json"))["tree_choices"]
    # a second, irregular tree: a parent whose children are all leaves sits BEFORE one with grandchildren ([0] vs [1])
    odd_tree = [[0], [1], [2], [0, 0], [0, 1], [1, 0], [1, 2], [2, 1], [1, 0, 0], [1, 0, 3], [1, 2, 1], [2, 1, 0], [1, 0, 0, 2]]
    out = {"cfg_keys": np.array(list(cfg_kw.keys())), "cfg_vals": np.array(list(cfg_kw.values()), dtype=np.int64), "rms_eps": 1e-6}
    for name, choices in (("std", tree), ("odd", odd_tree)):
        model = EagleModel(cfg, bias=True).float().eval()
        for p in model.parameters():
            torch.nn.init.normal_(p, std=0.08)
        model.layers[0].post_attention_layernorm.weight.data.fill_(1.0).add_(torch.randn(64) * 0.05)
        head = torch.nn.Linear(64, 320, bias=False)
        torch.nn.init.normal_(head.weight, std=2.5)
        plugin = object.__new__(Eagle)           # the constructor wants a checkpoint on disk; the methods under test do not
        torch.nn.Module.__init__(plugin) if isinstance(plugin, torch.nn.Module) else None
        plugin.tree, plugin.dtype, plugin.device, plugin.head, plugin.model = choices, torch.float32, "cpu", head, model
        plugin.accpet_tokens = plugin.accept_hidden_states = plugin.tree_indices = None
        model.gen_buffers(choices, "cpu")
        buf = plugin.gen_buffers()
        plugin.reset()
        out[f"{name}:choices"] = np.array(json.dumps(choices))
        out[f"{name}:head_weight"] = head.weight.detach().numpy()
        for k, v in model.state_dict().items():
            out[f"{name}:w:" + k] = v.numpy()
        out[f"{name}:mask"] = buf["tree_attn_mask"][0, 0].numpy().astype(np.uint8)
        out[f"{name}:pos"] = buf["tree_position_ids"].view(-1).numpy()
        out[f"{name}:retrieve"] = buf["tree_retrieve_indices"].numpy()
        out[f"{name}:tree_indices"] = buf["tree_indices"].numpy()
        rng = np.random.default_rng(11)
        steps = [13, 1, 4, 2, 6]                 # prompt, then accepted-token counts of later steps (the head's KV cache grows)
        for ci, t in enumerate(steps):
            # two update() calls per draft for ci == 2: the plugin concatenates what it is given until the next gen_draft
            parts = [t] if ci != 2 else [1, t - 1]
            hs_all, tok_all = [], []
            for n in parts:
                hs = torch.tensor(rng.normal(size=(n, 64)).astype(np.float32))
                toks = torch.tensor(rng.integers(3, 320, n))
                plugin.update(toks, hs)
                hs_all.append(hs); tok_all.append(toks)
            start = int(rng.integers(3, 320))
            with torch.no_grad():
                pred, _ = plugin.gen_draft(start)
            out[f"{name}:c{ci}:hidden"] = torch.cat(hs_all).numpy()
            out[f"{name}:c{ci}:tokens_in"] = torch.cat(tok_all).numpy()
            out[f"{name}:c{ci}:start"] = start
            out[f"{name}:c{ci}:draft"] = np.asarray(pred, dtype=np.int64)
            print(f"  eagle[{name}] call {ci}: T={t} draft[:8]={pred[:8]} n={len(pred)}")
        out[f"{name}:n_calls"] = len(steps)
    np.savez_compressed(os.path.join(HERE, "eagle.npz"), **out)
    print("wrote eagle.npz", os.path.getsize(os.path.join(HERE, "eagle.npz")), "bytes")


if __name__ == "__main__":
    main()
