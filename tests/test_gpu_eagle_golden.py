"""GPU parity of the EAGLE-2 / EAGLE (v1) tree expansion against outputs RECORDED FROM THE IMPORTED REFERENCE
(Eagle2Model.topk_genrate, samd/tree_model/eagle2/eagle2_model.py:820-975; Eagle.gen_draft over EagleModel.topk_genrate,
samd/tree_model/eagle/eagle.py:52-69, eagle_model.py:783-845).

Two layers:

 1. fp32 on the GPU (the recorded configuration is fp32): `Eagle2Head.topk_generate` / `Eagle.gen_draft` run on cuda:0 with the
    fixtures tests/golden/eagle2.npz / eagle.npz; draft tokens must be IDENTICAL, and the mask / positions / retrieve rows that the
    tree-buffer kernel (samd_tree_buffers, through the C ABI) derives from our parent array must equal the tensors the reference
    built with its host loops (eagle2_model.py:915-946; eagle/utils.py:62-212).

 2. the fp16 device head (samd/tree_model/device_head.py: the head on the library's gfx950 kernels -- streaming GEMM, RoPE + K/V
    write, tree attention) on the head_dim-128 fixtures eagle2_hd128.npz / eagle_hd128.npz (fp16-representable seeded weights,
    reference run in fp32).  Integer results must be identical wherever the reference's decision is decided by more than the
    fp16 tolerance below: every torch.topk call of the reference was recorded (values, indices, runner-up), the device path
    records the same sequence, and they are compared call by call.  Every recorded call must end in one of three ways:
      (a) identical: same tokens, same parent numbering, same mask / positions / retrieve rows;
      (b) EAGLE-2 only -- the same TREE (set of root->node token paths) under another node numbering: two near-equal candidates
          inside one row swapped ranks, which renumbers nodes but changes nothing the verify step can see;
      (c) a different draft whose FIRST differing top-k decision is a near-tie of the reference: recorded margin
          (value[p] - value[p+1]) <= 2 * TOL_FP16, with all values before it within TOL_FP16.
    Anything else fails, and at least MIN_SAME of a fixture's calls must end in (a) or (b): 80 % for the fp16 head (on the box every
    call of the V = 512 fixtures is (a)).  For the bf16 head on the RANDOM-weight fixture no share is demanded: bf16 carries 8x the
    rounding noise, the ~300 ordered decisions of an expansion over random weights include margins of 0.005-0.04, and on the box all
    five calls end in (c) at such a margin (the differing decision's recorded margin must be <= TIE_CAP_BF16 = 0.1).

 3. round 4: the bf16 device head on the PLANTED fixture (eagle2_planted_bf16.npz, V = 32000), where reproduction is decidable: every
    recorded decision has a margin >= 0.25 = 10x the measured bf16 noise, and the drafts must be identical (>= 80 % of the calls; all
    five on the box) -- test_eagle2_bf16_device_head_reproduces_reference_on_the_planted_fixture.

Fixtures: eagle2_hd128.npz / eagle_hd128.npz (fp16-representable weights, V = 512), eagle2_hd128_v32k.npz (V = 32000: the row
statistics take their split path) and eagle2_hd128_bf16.npz (weights and inputs representable in bf16 -- configs[3] computes in
bf16 -- followed by the bf16 device head).

TOL_FP16 = 0.1 absolute on head logits / log-probabilities.  The head's logits reach |x| ~ 64-128 here, where fp16 values are
0.0625 apart (the device head's logits are fp16 GEMM outputs), and its output states carry ~1e-3 relative error on top.
TOL_BF16 = 0.8 = 8 x TOL_FP16: bf16 keeps 3 mantissa bits fewer."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from eagle_fixture_weights import CFG, call_inputs, head_state, lm_head_weight

HERE = os.path.dirname(os.path.abspath(__file__))
TOL_FP16 = 0.1
TOL_BF16 = 0.8
TIE_CAP_BF16 = 0.1                 # random-weight bf16 fixture: a differing decision must be a recorded near-tie below this (observed 0.005-0.04)
NOISE_BF16_PLANTED = 0.04          # bound on |device value - reference value| of the planted fixture's top-k values (measured on MI355X: 0.0250);
                                   # accepted near-tie margin = 2x = 0.08, the fixture's smallest recorded margin is 0.25 = 10x the measured noise
MIN_SAME = {"f16": 0.8, "bf16": 0.0, "bf16_planted": 0.8}


def tree_buffers(parents):
    """mask [n,n] u8, positions [n], retrieve [leaves, depth] from the tree-buffer kernel (C ABI)"""
    n = len(parents)
    par = torch.tensor(parents, dtype=torch.int32, device="cuda")
    pos = torch.zeros(n, dtype=torch.int32, device="cuda")
    mb = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
    ret = torch.full((n * n,), -1, dtype=torch.int32, device="cuda")
    shape = torch.zeros(2, dtype=torch.int32, device="cuda")
    samd_hip.check(samd_hip.lib().samd_tree_buffers(samd_hip._ptr(par), n, 0, samd_hip._ptr(pos), None, samd_hip._ptr(mb), samd_hip._ptr(ret),
                                                    samd_hip._ptr(shape), samd_hip.current_stream()))
    nl, md = shape.cpu().tolist()
    return mb.cpu().numpy(), pos.cpu().numpy(), ret[:nl * md].reshape(nl, md).cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# 1. fp32 head on the GPU vs the fp32 recordings
# ---------------------------------------------------------------------------------------------------------------------
def test_eagle2_fp32_on_gpu_matches_recorded_reference():
    from samd.tree_model.eagle2 import Eagle2Head
    z = np.load(os.path.join(HERE, "golden", "eagle2.npz"))
    cfg = dict(zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()))
    cfg["rms_norm_eps"] = float(z["rms_eps"])
    head = Eagle2Head(cfg, dtype=torch.float32, device="cuda", bias=True)
    head.load_state({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    lm_head = torch.from_numpy(z["head_weight"]).cuda()
    head.reset()
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for ci in range(int(z["n_calls"])):
            toks, parents = head.topk_generate(torch.from_numpy(z[f"c{ci}:hidden"]).cuda(), torch.from_numpy(z[f"c{ci}:ids"]).cuda(), lm_head)
            assert toks.tolist() == z[f"c{ci}:tokens"].tolist(), f"draft tokens differ in call {ci}"
            par = parents.tolist()
            assert par[0] == -1 and all(0 <= par[i] < i for i in range(1, len(par)))
            mask, pos, ret = tree_buffers(par)
            assert mask.tolist() == z[f"c{ci}:mask"].tolist()
            assert pos.tolist() == z[f"c{ci}:pos"].tolist()
            assert ret.tolist() == z[f"c{ci}:retrieve"].tolist()
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev


@pytest.mark.parametrize("name", ["std", "odd"])
def test_eagle_v1_fp32_on_gpu_matches_recorded_reference(name):
    from samd.tree_model.eagle import Eagle, EagleHead, StaticDraftTree
    z = np.load(os.path.join(HERE, "golden", "eagle.npz"))
    cfg = dict(zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()))
    cfg["rms_norm_eps"] = float(z["rms_eps"])
    head = EagleHead(cfg, dtype=torch.float32, device="cuda", bias=True)
    pre = f"{name}:w:"
    head.load_state({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    choices = json.loads(str(z[f"{name}:choices"]))
    head.set_tree(StaticDraftTree(choices))

    class LM:
        lm_head = torch.nn.Linear(64, 320, bias=False).cuda()
    LM.lm_head.weight.data = torch.from_numpy(z[f"{name}:head_weight"]).cuda()
    plug = Eagle(None, LM, torch.float32, "cuda", head=head)
    assert plug.device_head is None                      # head_dim 16: the PyTorch forward, in fp32, on the GPU
    # static buffers: the tree-buffer kernel on our parent array vs the reference's gen_buffers (eagle/utils.py:62-212)
    mask, pos, ret = tree_buffers(plug.tree.parents)
    assert mask.tolist() == z[f"{name}:mask"].tolist() and pos.tolist() == z[f"{name}:pos"].tolist()
    assert sorted(map(tuple, ret.tolist())) == sorted(map(tuple, z[f"{name}:retrieve"].tolist()))   # same paths; the reference sorts its rows
    buf = plug.gen_buffers()
    assert buf["tree_retrieve_indices"].tolist() == z[f"{name}:retrieve"].tolist() and buf["tree_indices"].tolist() == z[f"{name}:tree_indices"].tolist()
    plug.reset()
    for ci in range(int(z[f"{name}:n_calls"])):
        hs, toks = torch.from_numpy(z[f"{name}:c{ci}:hidden"]).cuda(), torch.from_numpy(z[f"{name}:c{ci}:tokens_in"]).cuda()
        plug.update(tokens=toks, last_hidden_states=hs)
        draft, kwargs = plug.gen_draft(int(z[f"{name}:c{ci}:start"]))
        assert kwargs == {} and draft == z[f"{name}:c{ci}:draft"].tolist(), f"draft tokens differ in call {ci}"


# ---------------------------------------------------------------------------------------------------------------------
# 2. the fp16 device head vs the head_dim-128 recordings, decision by decision
# ---------------------------------------------------------------------------------------------------------------------
def follow(ref_calls, dev_calls, last_is_a_set=False, tol=TOL_FP16, tie_cap=None):
    """compare two top-k decision sequences -> None when every call agrees (indices equal, values within TOL_FP16), else
    (call j, row, position, recorded margin) of the first difference, which must be a near-tie of the reference.
    last_is_a_set: the final call only selects (EAGLE-2 sorts the kept candidates by index afterwards, eagle2_model.py:893-895)."""
    assert len(ref_calls) == len(dev_calls)
    tie_cap = 2 * tol if tie_cap is None else tie_cap            # the largest recorded margin a differing decision may have
    for j, ((rv, ri, rnext), (dv, di)) in enumerate(zip(ref_calls, dev_calls)):
        rv2, ri2, dv2, di2 = (np.asarray(a).reshape(-1, np.asarray(a).shape[-1]) for a in (rv, ri, dv, di))
        if last_is_a_set and j == len(ref_calls) - 1 and sorted(ri2.reshape(-1).tolist()) == sorted(di2.reshape(-1).tolist()):
            continue
        rn = np.asarray(rnext, dtype=np.float32).reshape(-1)
        for row in range(rv2.shape[0]):
            ext = np.append(rv2[row], rn[row])
            for p in range(rv2.shape[1]):
                if ri2[row, p] != di2[row, p]:
                    margin = float(ext[p] - ext[p + 1])
                    assert margin <= tie_cap, (f"top-k call {j} row {row} position {p}: device chose {di2[row, p]}, reference {ri2[row, p]} "
                                               f"with margin {margin:.4f} > {tie_cap}")
                    return j, row, p, margin
                assert abs(float(rv2[row, p]) - float(dv2[row, p])) <= tol, (j, row, p, float(rv2[row, p]), float(dv2[row, p]))
    return None


def ref_trace(z, prefix):
    return [(z[f"{prefix}:k{j}:v"], z[f"{prefix}:k{j}:i"], z[f"{prefix}:k{j}:next"]) for j in range(int(z[f"{prefix}:n_topk"]))]


def device_head_for(seed, head_cls, dtype=torch.float16, rounding="f16", vocab=512, state=None, lm_head=None):
    """the seeded head in `dtype` on a LlamaRunner whose lm_head is the fixture's (the base model's layers are irrelevant here).
    state / lm_head: explicit weights (the planted fixture) instead of the seeded random ones"""
    from samd_hip.llama import LlamaRunner
    from samd.tree_model.device_head import DeviceHead
    cfg = dict(hidden_size=CFG["hidden_size"], intermediate_size=CFG["intermediate_size"], num_attention_heads=CFG["num_attention_heads"],
               num_key_value_heads=CFG["num_key_value_heads"], vocab_size=vocab, rms_norm_eps=CFG["rms_norm_eps"], rope_theta=10000.0)
    head = head_cls(cfg, dtype=dtype, device="cuda", bias=True)
    head.load_state({k: torch.from_numpy(v) for k, v in (state if state is not None else head_state(seed, rounding, vocab)).items()})
    base_cfg = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, head_dim=128,
                    vocab_size=vocab, max_position_embeddings=512, rms_norm_eps=1e-6)
    runner = LlamaRunner.random_init(base_cfg, 256, dtype, seed=1)
    runner.w["lm_head"].copy_(torch.from_numpy(lm_head if lm_head is not None else lm_head_weight(seed, rounding=rounding, vocab=vocab)).to(dtype))
    samd_hip.check(samd_hip.lib().samd_gemm_pack_weights(samd_hip._ptr(runner.w["lm_head"]), samd_hip._ptr(runner.wp["lm_head"]), vocab, 256,
                                                         samd_hip.current_stream()))
    torch.cuda.synchronize()
    return head, runner, DeviceHead(head, runner)


def path_set(tokens, mask, pos):
    """the draft as a set of root->node token paths: what verification sees, whatever the node numbering"""
    out = set()
    for i in range(len(tokens)):
        anc = sorted(np.nonzero(np.asarray(mask)[i])[0].tolist(), key=lambda j: int(pos[j]))
        out.add(tuple(int(tokens[j]) for j in anc))
    return out


@pytest.mark.parametrize("fixture,kind", [("eagle2_hd128.npz", "f16"), ("eagle2_hd128_v32k.npz", "f16"), ("eagle2_hd128_bf16.npz", "bf16")])
def test_eagle2_device_head_follows_recorded_reference(fixture, kind):
    from samd.tree_model.eagle2 import Eagle2Head
    z = np.load(os.path.join(HERE, "golden", fixture))
    seed = int(z["seed"])
    vocab = int(z["vocab"]) if "vocab" in z.files else 512
    dtype, tol = (torch.float16, TOL_FP16) if kind == "f16" else (torch.bfloat16, TOL_BF16)
    head, runner, dh = device_head_for(seed, Eagle2Head, dtype, kind, vocab)
    dh.reset()
    good, notes = 0, []
    n_calls = len(z["steps"].tolist())
    for ci, t in enumerate(z["steps"].tolist()):
        hs, ids = call_inputs(seed, ci, t, kind, vocab)
        head.trace = []
        toks, parents = head.topk_generate_device(dh, torch.from_numpy(hs).cuda().to(dtype), torch.from_numpy(ids).cuda())
        torch.cuda.synchronize()
        dev_calls, head.trace = head.trace, None
        mask, pos, ret = tree_buffers(parents.tolist())
        if (toks.tolist() == z[f"c{ci}:tokens"].tolist() and mask.tolist() == z[f"c{ci}:mask"].tolist() and pos.tolist() == z[f"c{ci}:pos"].tolist()
                and ret.tolist() == z[f"c{ci}:retrieve"].tolist()):
            notes.append(f"call {ci}: identical")
            good += 1
        elif path_set(toks.tolist(), mask, pos) == path_set(z[f"c{ci}:tokens"], z[f"c{ci}:mask"], z[f"c{ci}:pos"]):
            notes.append(f"call {ci}: same tree, other numbering")
            good += 1
        else:
            # asserts that the first difference is a near-tie; bf16: values may sit 0.8 apart (logits of 64-128 are 0.5 apart in bf16),
            # but the DECISION that differs must be closer than TIE_CAP_BF16 = 0.1 in the reference (observed: 0.005-0.04)
            where = follow(ref_trace(z, f"c{ci}"), dev_calls, last_is_a_set=True, tol=tol, tie_cap=TIE_CAP_BF16 if kind == "bf16" else None)
            assert where is not None, f"call {ci}: drafts differ although every recorded decision matches"
            notes.append(f"call {ci}: near-tie at top-k call {where[0]} row {where[1]} pos {where[2]} (recorded margin {where[3]:.4f})")
    summary = f"eagle2 {kind} device head vs recorded reference ({fixture}): " + "; ".join(notes)
    print(summary)
    # every call ended in (a), (b) or a proven near-tie (anything else asserted above); most must be the same draft
    assert good >= MIN_SAME[kind] * n_calls, summary


def test_eagle2_bf16_device_head_reproduces_reference_on_the_planted_fixture():
    """configs[3]'s dtype on a fixture where reproduction is DECIDABLE (tests/golden/eagle2_planted_bf16.npz, generator
    make_golden_eagle_planted.py): V = 32000, head_dim 128, every weight and input representable in bf16, and each of the reference's
    1845 ordered top-k decisions (five calls, fp32 on the CPU) recorded with a margin >= 0.25 -- asserted at fixture time and
    re-asserted here from the recorded values.  The bf16 device head must produce the IDENTICAL draft (tokens, mask, positions,
    retrieve rows) in >= 80 % of the calls; a call that differs must still be a proven near-tie of the reference with margin
    <= 2 x NOISE_BF16_PLANTED = 0.1 -- which this fixture does not contain, so any real difference fails.  Every value the device
    records next to an identical decision must lie within NOISE_BF16_PLANTED of the reference's (measured on MI355X: see the printed
    summary; bf16 logits of magnitude 2-4 are 0.0156 apart, log-softmax and cumulative scores are fp32 on the device)."""
    from samd.tree_model.eagle2 import Eagle2Head
    import eagle_fixture_weights as W
    z = np.load(os.path.join(HERE, "golden", "eagle2_planted_bf16.npz"))
    seed, vocab, margin = int(z["seed"]), int(z["vocab"]), float(z["margin"])
    assert vocab >= 32000 and margin > 3 * 2 * NOISE_BF16_PLANTED
    n_calls = len(z["steps"].tolist())
    # the recorded margins themselves: every kept position against the next value, per top-k call (the last call only selects a set)
    decisions = 0
    for ci in range(n_calls):
        calls = ref_trace(z, f"c{ci}")
        for j, (rv, ri, rnext) in enumerate(calls):
            rv2 = np.asarray(rv).reshape(-1, np.asarray(rv).shape[-1])
            ext = np.concatenate([rv2, np.asarray(rnext, dtype=np.float32).reshape(-1, 1)], axis=1)
            gaps = ext[:, :-1] - ext[:, 1:]
            if j == len(calls) - 1:
                gaps = gaps[:, -1:]
            assert gaps.min() >= margin, (ci, j, float(gaps.min()))
            decisions += gaps.size
    state = W.planted_head_state(seed, z["slot_token"])
    lm = W.planted_lm_head(seed, z["slot_succ"], z["slot_logit"])
    head, runner, dh = device_head_for(seed, Eagle2Head, torch.bfloat16, "bf16", vocab, state=state, lm_head=lm)
    dh.reset()
    good, notes, noise = 0, [], 0.0
    for ci, t in enumerate(z["steps"].tolist()):
        hs, ids = W.planted_call_inputs(seed, ci, t, int(z["roots"][ci]))
        head.trace = []
        toks, parents = head.topk_generate_device(dh, torch.from_numpy(hs).cuda().to(torch.bfloat16), torch.from_numpy(ids).cuda())
        torch.cuda.synchronize()
        dev_calls, head.trace = head.trace, None
        mask, pos, ret = tree_buffers(parents.tolist())
        where = follow(ref_trace(z, f"c{ci}"), dev_calls, last_is_a_set=True, tol=NOISE_BF16_PLANTED)
        for (rv, ri, _), (dv, di) in zip(ref_trace(z, f"c{ci}"), dev_calls):
            if np.array_equal(np.asarray(ri).reshape(-1), np.asarray(di).reshape(-1)):
                noise = max(noise, float(np.abs(np.asarray(rv, dtype=np.float32).reshape(-1) - np.asarray(dv, dtype=np.float32).reshape(-1)).max()))
        if (toks.tolist() == z[f"c{ci}:tokens"].tolist() and mask.tolist() == z[f"c{ci}:mask"].tolist() and pos.tolist() == z[f"c{ci}:pos"].tolist()
                and ret.tolist() == z[f"c{ci}:retrieve"].tolist()):
            assert where is None
            notes.append(f"call {ci}: identical")
            good += 1
        else:
            assert where is not None, f"call {ci}: drafts differ although every recorded decision matches"
            notes.append(f"call {ci}: near-tie at top-k call {where[0]} row {where[1]} pos {where[2]} (recorded margin {where[3]:.4f})")
    summary = (f"eagle2 bf16 device head on the planted fixture ({decisions} recorded decisions, every margin >= {margin}): " + "; ".join(notes)
               + f"; max |device - reference| on the followed values {noise:.4f}")
    print(summary)
    assert noise <= NOISE_BF16_PLANTED, summary
    assert good >= MIN_SAME["bf16_planted"] * n_calls, summary


@pytest.mark.parametrize("name", ["std", "odd"])
def test_eagle_v1_device_head_follows_recorded_reference(name):
    from samd.tree_model.eagle import Eagle, EagleHead, StaticDraftTree
    z = np.load(os.path.join(HERE, "golden", "eagle_hd128.npz"))
    seed = int(z[f"{name}:seed"])
    head, runner, _ = device_head_for(seed, EagleHead)
    head.set_tree(StaticDraftTree(json.loads(str(z[f"{name}:choices"]))))
    plug = Eagle(None, runner, torch.float16, "cuda", head=head)
    assert plug.device_head is not None
    plug.reset()
    exact, notes = 0, []
    for ci, t in enumerate(z["steps"].tolist()):
        hs, ids = call_inputs(seed, ci, t)
        plug.update(tokens=torch.from_numpy(ids[:t]).cuda(), last_hidden_states=torch.from_numpy(hs).cuda().half())
        head.trace = []
        draft, _ = plug.gen_draft(int(ids[t]))
        torch.cuda.synchronize()
        dev_calls, head.trace = head.trace, None
        if draft == z[f"{name}:c{ci}:draft"].tolist():
            notes.append(f"call {ci}: identical")
            exact += 1
            continue
        where = follow(ref_trace(z, f"{name}:c{ci}"), dev_calls)                          # asserts that the first difference is a near-tie
        assert where is not None, f"call {ci}: drafts differ although every recorded decision matches"
        notes.append(f"call {ci}: near-tie at top-k call {where[0]} row {where[1]} pos {where[2]} (recorded margin {where[3]:.4f})")
    summary = f"eagle v1 [{name}] device head vs recorded reference: " + "; ".join(notes)
    print(summary)
    assert exact >= MIN_SAME["f16"] * len(z["steps"].tolist()), summary
