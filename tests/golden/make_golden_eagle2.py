"""Generates tests/golden/eagle2.npz by running the IMPORTED reference Eagle2Model.topk_genrate
(samd/tree_model/eagle2/eagle2_model.py:820-975) on a tiny random configuration, CPU, fp32.
Dev-container only (needs /root/reference); the fixture holds weights, inputs and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eagle2.py
"""
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

from samd.tree_model.eagle2.eagle2_config import Eagle2Config          # noqa: E402
from samd.tree_model.eagle2.eagle2_model import Eagle2Model            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(1234)
    cfg_kw = dict(vocab_size=320, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-6, pad_token_id=0)
    cfg = Eagle2Config(**cfg_kw)
    cfg.rope_scaling = None                      # transformers 5.x injects a default dict the 2023 code cannot parse
    model = Eagle2Model(cfg, bias=True).float().eval()
    for p in model.parameters():
        torch.nn.init.normal_(p, std=0.08)
    model.layers[0].post_attention_layernorm.weight.data.fill_(1.0).add_(torch.randn(64) * 0.05)
    model.init_tree()
    model.stable_kv = None
    head = torch.nn.Linear(64, 320, bias=False)
    torch.nn.init.normal_(head.weight, std=2.5)
    out = {"cfg_keys": np.array(list(cfg_kw.keys())), "cfg_vals": np.array(list(cfg_kw.values()), dtype=np.int64), "rms_eps": 1e-6,
           "head_weight": head.weight.detach().numpy()}
    for k, v in model.state_dict().items():
        out["w:" + k] = v.numpy()
    rng = np.random.default_rng(7)
    calls = []
    for ci, t in enumerate([13, 1, 4, 2, 9]):     # prompt, then accepted-token counts of later steps (the head's KV cache grows)
        hs = torch.tensor(rng.normal(size=(t, 64)).astype(np.float32))
        ids = torch.tensor(rng.integers(3, 320, t + 1))
        with torch.no_grad():
            toks, buf = model.topk_genrate(hs, ids, head)
        out[f"c{ci}:hidden"] = hs.numpy()
        out[f"c{ci}:ids"] = ids.numpy()
        out[f"c{ci}:tokens"] = toks.view(-1).numpy()
        out[f"c{ci}:mask"] = buf["tree_attn_mask"][0, 0].numpy().astype(np.uint8)
        out[f"c{ci}:pos"] = buf["tree_position_ids"].view(-1).numpy()
        out[f"c{ci}:retrieve"] = buf["tree_retrieve_indices"].numpy()
        calls.append(t)
        print(f"  eagle2 call {ci}: T={t} tokens[:6]={toks.view(-1)[:6].tolist()} leaves={buf['tree_retrieve_indices'].shape}")
    out["n_calls"] = len(calls)
    np.savez_compressed(os.path.join(HERE, "eagle2.npz"), **out)
    print("wrote eagle2.npz", os.path.getsize(os.path.join(HERE, "eagle2.npz")), "bytes")


if __name__ == "__main__":
    main()
