"""Generates tests/golden/eagle2_hd128.npz and eagle_hd128.npz: the IMPORTED reference's EAGLE-2 expansion
(Eagle2Model.topk_genrate, samd/tree_model/eagle2/eagle2_model.py:820-975) and EAGLE v1 plugin (Eagle.update / gen_draft,
samd/tree_model/eagle/eagle.py:13-75) on a head_dim-128 configuration (hidden 256, 2 heads), CPU fp32, with fp16-representable
seeded weights (tests/eagle_fixture_weights.py) -- the shape the gfx950 device head (samd/tree_model/device_head.py) runs, so the
GPU tests can compare its fp16 path with the reference's recorded drafts.

Half precision moves logits by a few 1e-4 relative, which can swap two near-equal candidates, and with ~300 ordered top-k
decisions per expansion over random weights some decision is always close.  So (a) every torch.topk call of the reference is
recorded -- values, indices and the runner-up value (the margin of the last kept entry) -- which lets the GPU test follow the
device head decision by decision and accept a difference only where the reference's own margin is below the stated fp16
tolerance; (b) the seed is the first one (of 400; else the most robust one seen -- EAGLE-2: seed 86 survives 2 of 8) whose drafts survive 8 runs of the reference with 2e-3 relative noise on its
hidden states and head logits -- for EAGLE-2 as a TREE (the set of root->node token paths: a swap of two near-equal
candidates inside a row renumbers nodes without changing the tree), for EAGLE v1 exactly (there the rank decides which child
slot a token fills) -- so that most recorded calls are expected to match.  Dev-container only (needs
/root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eagle_hd128.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, R)
sys.path.insert(0, os.path.dirname(HERE))
for pkg in ("samd_sam_only", "samd"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{R}/{pkg}"]
    sys.modules[pkg] = m

from samd.tree_model.eagle2.eagle2_config import Eagle2Config          # noqa: E402
from samd.tree_model.eagle2.eagle2_model import Eagle2Model            # noqa: E402
from samd.tree_model.eagle.eagle_config import EagleConfig              # noqa: E402
from samd.tree_model.eagle.eagle_model import EagleModel                # noqa: E402
from samd.tree_model.eagle.eagle import Eagle                           # noqa: E402
from eagle_fixture_weights import CFG, call_inputs, head_state, lm_head_weight   # noqa: E402

SEEDS = dict(eagle2=86, std=15, odd=7)    # results of the search below, so that regenerating takes seconds (delete an entry to search again)
STEPS = [13, 1, 4]          # prompt, then accepted-token counts of later steps (the head's KV cache grows)
# round 3: the same recording for config 4's dtype and for a real vocabulary size
#   eagle2_hd128_bf16.npz : weights / inputs representable in bf16 (the Llama-3-8B configuration computes in bf16)
#   eagle2_hd128_v32k.npz : V = 32000 (Vicuna's vocabulary: the row statistics run their split path, 8 workgroups per row)
# seed = result of the search (most (call, noisy trial) pairs that keep the recorded tree); None = search again
VARIANTS = dict(bf16=dict(rounding="bf16", vocab=512, steps=[13, 1, 4, 2, 3], seed=None, limit=200, noise=4e-3),
                v32k=dict(rounding="f16", vocab=32000, steps=[13, 1, 4, 2, 3], seed=None, limit=40, noise=2e-3))
NOISE, TRIALS = 2e-3, 8


class TopkTrace:
    """records every torch.topk call made inside the block: (values, indices, runner-up value per row)"""

    def __enter__(self):
        self.calls, self.orig = [], torch.topk
        torch.topk = self.hook
        return self

    def __exit__(self, *exc):
        torch.topk = self.orig

    def hook(self, x, k, dim=-1, **kw):
        r = self.orig(x, k, dim=dim, **kw)
        n = x.shape[dim]
        nxt = self.orig(x, k + 1, dim=dim).values.select(dim, k) if n > k else torch.full(x.shape[:-1], float("-inf"))
        self.calls.append((r.values.detach().numpy().astype(np.float32).copy(), r.indices.detach().numpy().copy(),
                           np.asarray(nxt.detach().numpy(), dtype=np.float32).copy()))
        return r


class NoisyHead:
    """lm_head with relative noise on its input and output (noise 0 = the plain linear map)"""

    def __init__(self, w, noise, gen):
        self.w, self.noise, self.gen = w, noise, gen

    def __call__(self, h):
        if self.noise:
            h = h * (1 + self.noise * torch.randn(h.shape, generator=self.gen))
        y = torch.nn.functional.linear(h, self.w)
        if self.noise:
            y = y * (1 + self.noise * torch.randn(y.shape, generator=self.gen))
        return y


def eagle2_run(seed, noise=0.0, trial=0, want_trace=False, rounding="f16", vocab=512, steps=None):
    cfg = Eagle2Config(**dict(CFG, vocab_size=vocab))
    cfg.rope_scaling = None
    model = Eagle2Model(cfg, bias=True).float().eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in head_state(seed, rounding, vocab).items()}, strict=True)
    model.init_tree()
    model.stable_kv = None
    gen = torch.Generator().manual_seed(1000 * seed + trial)
    head = NoisyHead(torch.from_numpy(lm_head_weight(seed, rounding=rounding, vocab=vocab)), noise, gen)
    outs, traces = [], []
    for ci, t in enumerate(steps or STEPS):
        hs, ids = call_inputs(seed, ci, t, rounding, vocab)
        hs = torch.from_numpy(hs)
        if noise:
            hs = hs * (1 + noise * torch.randn(hs.shape, generator=gen))
        with torch.no_grad(), TopkTrace() as tr:
            toks, buf = model.topk_genrate(hs, torch.from_numpy(ids), head)
        traces.append(tr.calls)
        outs.append((toks.view(-1).numpy().copy(), buf["tree_attn_mask"][0, 0].numpy().astype(np.uint8),
                     buf["tree_position_ids"].view(-1).numpy().copy(), buf["tree_retrieve_indices"].numpy().copy()))
    return (outs, traces) if want_trace else outs


def eagle_run(seed, choices, noise=0.0, trial=0, want_trace=False):
    cfg = EagleConfig(**CFG)
    cfg.rope_scaling = None
    model = EagleModel(cfg, bias=True).float().eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in head_state(seed).items()}, strict=True)
    gen = torch.Generator().manual_seed(2000 * seed + trial)
    head = NoisyHead(torch.from_numpy(lm_head_weight(seed)), noise, gen)
    plugin = object.__new__(Eagle)           # the constructor wants a checkpoint on disk; the methods under test do not
    if isinstance(plugin, torch.nn.Module):
        torch.nn.Module.__init__(plugin)
    plugin.tree, plugin.dtype, plugin.device, plugin.head, plugin.model = choices, torch.float32, "cpu", head, model
    plugin.accpet_tokens = plugin.accept_hidden_states = plugin.tree_indices = None
    model.gen_buffers(choices, "cpu")
    buf = plugin.gen_buffers()
    plugin.reset()
    outs, traces = [], []
    for ci, t in enumerate(STEPS):
        hs, ids = call_inputs(seed, ci, t)
        hs = torch.from_numpy(hs)
        if noise:
            hs = hs * (1 + noise * torch.randn(hs.shape, generator=gen))
        plugin.update(torch.from_numpy(ids[:t]), hs)
        with torch.no_grad(), TopkTrace() as tr:
            pred, _ = plugin.gen_draft(int(ids[t]))
        traces.append(tr.calls)
        outs.append(np.asarray(pred, dtype=np.int64))
    return (outs, buf, traces) if want_trace else (outs, buf)


def path_set(toks, mask, pos):
    """the draft as a set of root->node token paths: what verification sees, whatever the node numbering"""
    out = set()
    for i in range(len(toks)):
        anc = sorted(np.nonzero(mask[i])[0].tolist(), key=lambda j: int(pos[j]))
        out.add(tuple(int(toks[j]) for j in anc))
    return out


def same_tree(a, b):
    return all(path_set(ca[0], ca[1], ca[2]) == path_set(cb[0], cb[1], cb[2]) for ca, cb in zip(a, b))


def same(a, b):
    return all(all(np.array_equal(x, y) for x, y in zip(ca, cb)) if isinstance(ca, tuple) else np.array_equal(ca, cb) for ca, cb in zip(a, b))


def pick_seed(run, limit=400, known=None, same=same):
    """first seed whose outputs survive TRIALS noisy runs of the reference; else the most robust one seen.  `known` = the seed
    an earlier run of this search found (SEEDS below): it is re-checked instead of searched for."""
    best = (-1, None)
    for seed in ([known] if known else range(1, limit)):
        base = run(seed, 0.0, 0)
        ok = sum(same(base, run(seed, NOISE, t)) for t in range(TRIALS))
        if ok > best[0]:
            best = (ok, seed)
        if ok == TRIALS:
            break
    return best[1], best[0]


def put_trace(out, prefix, calls):
    out[f"{prefix}:n_topk"] = len(calls)
    for j, (v, i, nxt) in enumerate(calls):
        out[f"{prefix}:k{j}:v"], out[f"{prefix}:k{j}:i"], out[f"{prefix}:k{j}:next"] = v, i, nxt


def eagle2_variant(name, rounding, vocab, steps, seed, limit, noise):
    """seed = the one whose recorded trees survive the most (call, noisy trial) pairs: with five calls per fixture hardly any seed
    keeps ALL of them under noise, but the GPU test counts calls"""
    run = lambda s, n, t: eagle2_run(s, n, t, rounding=rounding, vocab=vocab, steps=steps)
    best = (-1, None)
    for cand in ([seed] if seed else range(1, limit)):
        base = run(cand, 0.0, 0)
        ok = sum(path_set(a[0], a[1], a[2]) == path_set(b[0], b[1], b[2]) for t in range(TRIALS) for a, b in zip(base, run(cand, noise, t)))
        if ok > best[0]:
            best = (ok, cand)
            print(f"    {name}: seed {cand} keeps {ok}/{TRIALS * len(steps)} (call, trial) pairs", flush=True)
        if ok == TRIALS * len(steps):
            break
    ok, seed = best
    base, traces = eagle2_run(seed, want_trace=True, rounding=rounding, vocab=vocab, steps=steps)
    out = {"seed": seed, "steps": np.array(steps), "noise": noise, "trials": TRIALS, "robust_pairs": ok, "vocab": vocab,
           "rounding": np.array(rounding)}
    for ci, (toks, mask, pos, ret) in enumerate(base):
        out[f"c{ci}:tokens"], out[f"c{ci}:mask"], out[f"c{ci}:pos"], out[f"c{ci}:retrieve"] = toks, mask, pos, ret
        put_trace(out, f"c{ci}", traces[ci])
        print(f"  eagle2_hd128_{name} seed {seed} ({ok}/{TRIALS * len(steps)} (call, noisy trial) pairs keep the tree) call {ci}: tokens[:6]={toks[:6].tolist()} leaves={ret.shape}")
    f = os.path.join(HERE, f"eagle2_hd128_{name}.npz")
    np.savez_compressed(f, **out)
    print("wrote", f, os.path.getsize(f), "bytes")


def main():
    if len(sys.argv) > 1:                     # only the named round-3 variants: python make_golden_eagle_hd128.py bf16 v32k
        for name in sys.argv[1:]:
            eagle2_variant(name, **VARIANTS[name])
        return
    # ---- EAGLE-2 ------------------------------------------------------------------------------------------------
    seed, ok = pick_seed(lambda s, n, t: eagle2_run(s, n, t), known=SEEDS.get("eagle2"), same=same_tree)
    base, traces = eagle2_run(seed, want_trace=True)
    out = {"seed": seed, "steps": np.array(STEPS), "noise": NOISE, "trials": TRIALS, "robust_trials": ok}
    for ci, (toks, mask, pos, ret) in enumerate(base):
        out[f"c{ci}:tokens"], out[f"c{ci}:mask"], out[f"c{ci}:pos"], out[f"c{ci}:retrieve"] = toks, mask, pos, ret
        put_trace(out, f"c{ci}", traces[ci])
        print(f"  eagle2_hd128 seed {seed} ({ok}/{TRIALS} noisy runs identical) call {ci}: tokens[:6]={toks[:6].tolist()} leaves={ret.shape} topk calls={len(traces[ci])}")
    np.savez_compressed(os.path.join(HERE, "eagle2_hd128.npz"), **out)
    # ---- EAGLE v1 (shipped tree + the irregular one of make_golden_eagle.py) -------------------------------------------
    tree = json.load(open(f"{R}/samd/config/eagle.json"))["tree_choices"]
    odd_tree = [[0], [1], [2], [0, 0], [0, 1], [1, 0], [1, 2], [2, 1], [1, 0, 0], [1, 0, 3], [1, 2, 1], [2, 1, 0], [1, 0, 0, 2]]
    out = {"steps": np.array(STEPS), "noise": NOISE, "trials": TRIALS}
    for name, choices in (("std", tree), ("odd", odd_tree)):
        seed, ok = pick_seed(lambda s, n, t: eagle_run(s, choices, n, t)[0], known=SEEDS.get(name))
        base, buf, traces = eagle_run(seed, choices, want_trace=True)
        out[f"{name}:seed"], out[f"{name}:robust_trials"] = seed, ok
        out[f"{name}:choices"] = np.array(json.dumps(choices))
        for ci, pred in enumerate(base):
            out[f"{name}:c{ci}:draft"] = pred
            put_trace(out, f"{name}:c{ci}", traces[ci])
            print(f"  eagle_hd128[{name}] seed {seed} ({ok}/{TRIALS}) call {ci}: draft[:8]={pred[:8].tolist()} n={len(pred)} topk calls={len(traces[ci])}")
    np.savez_compressed(os.path.join(HERE, "eagle_hd128.npz"), **out)
    for f in ("eagle2_hd128.npz", "eagle_hd128.npz"):
        print("wrote", f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
