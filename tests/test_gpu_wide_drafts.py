"""Drafts of 65-128 nodes on the GPU (round 5: SAMD_MAX_DRAFT 128; VERDICT r04 #6 -- the reference takes any max_predicts / n_predicts,
SO/sam/static_sam.py:183, S/sam/dyn_sam.py:107-113).  The session kernel's part -- drafts, buffers, accept, whole-loop traces at
max_predicts 80 / 100 / 128 and n_predicts 100 / 128 -- is followed through the SAME tests as the <= 64-node fixtures: tests/conftest.py
appends tests/golden/wide_*.json.gz (recorded from the imported reference by tests/golden/make_golden_wide.py) to every fixture.  This file
covers what those do not: the tree attention over two 64-row tiles with two mask words per row, the 128-row verify forward against
fp32 HuggingFace, and generate() at max_predicts 128 on a real decoder -- speculative output == autoregressive output."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from oracle import sam_oracle as O
from test_gpu_verify import reference_attention
from util import random_parents


def dev(a, dtype=torch.int32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def mask_words(anc, n):
    """(python-int rows, device [2][MAX_DRAFT] int64 array: low words then high words)"""
    m = O.gen_buffers(anc)["tree_attn_mask"][0, 0]
    rows = [sum(1 << j for j in range(n) if m[i, j]) for i in range(n)]
    M = samd_hip.MAX_DRAFT
    lo = [r & ((1 << 64) - 1) for r in rows] + [0] * (M - n)
    hi = [r >> 64 for r in rows] + [0] * (M - n)
    return rows, torch.tensor(np.array(lo + hi, dtype=np.uint64).view(np.int64), device="cuda")


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H,Hkv,L,n,shape", [(32, 32, 700, 128, "bushy"), (32, 8, 1500, 100, "random"), (8, 8, 0, 65, "chain"), (4, 2, 37, 127, "star"),
                                             (32, 32, 1900, 96, "chain"), (16, 16, 64, 128, "random")])
def test_tree_attention_two_row_tiles(dtype, tol, H, Hkv, L, n, shape):
    rng = np.random.default_rng(L * 13 + n)
    D, max_len, n_pad = 128, 2048, 128
    g = torch.Generator(device="cuda").manual_seed(L + n)
    q = torch.randn((n_pad, H, D), generator=g, device="cuda").to(dtype)
    k_cache = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    v_cache = (torch.randn((Hkv, max_len, D), generator=g, device="cuda") * torch.linspace(0.5, 2.0, D, device="cuda")).to(dtype)
    k_cache[:, L + n:] = float("nan")
    v_cache[:, L + n:] = float("nan")
    q[n:] = float("nan")
    anc = random_parents(rng, n, shape)
    rows, mask = mask_words(anc, n)
    out = torch.full((n_pad, H, D), 7.0, device="cuda").to(dtype)
    ws_bytes = samd_hip.lib().samd_tree_attention_workspace(n_pad, H, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    scale = 1.0 / math.sqrt(D)
    d_L, d_n = dev([L]), dev([n])
    samd_hip.check(samd_hip.lib().samd_tree_attention(samd_hip._ptr(q), samd_hip._ptr(k_cache), samd_hip._ptr(v_cache), samd_hip._ptr(out),
                                                      samd_hip.torch_dtype_code(dtype), n_pad, H, Hkv, D, max_len, samd_hip._ptr(mask),
                                                      samd_hip._ptr(d_L), samd_hip._ptr(d_n), scale, samd_hip._ptr(ws), ws_bytes, samd_hip.current_stream()))
    want = reference_attention(q, k_cache, v_cache, L, n, rows, scale)
    got = out[:n].float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err
    assert (out[n:] == 0).all()


def test_session_masks_of_a_128_node_tree_reach_the_host_and_the_kernel():
    """set_draft -> build_buffers (two nodes per lane) -> DraftHost.mask / mask_hi, positions and retrieve rows == the oracle's gen_buffers
    (the device layout the attention reads -- low words of all rows, then the high words -- is exercised by the 128-row forward below)"""
    rng = np.random.default_rng(5)
    sess = samd_hip.Session(512)
    for n, shape in ((128, "bushy"), (97, "random"), (65, "chain"), (128, "star"), (64, "random"), (3, "chain")):
        anc = random_parents(rng, n, shape)
        toks = rng.integers(3, 90, n).tolist()
        sess.set_draft(dev(toks), dev(anc), n, type_=1)
        d = sess.read_draft()
        b = O.gen_buffers(anc)
        want = b["tree_attn_mask"][0, 0]
        rows = [sum(1 << j for j in range(n) if want[i, j]) for i in range(n)]
        assert [int(x) for x in d.mask[:n]] == [r & ((1 << 64) - 1) for r in rows]
        assert [int(x) for x in d.mask_hi[:n]] == [r >> 64 for r in rows]
        assert all(int(x) == 0 for x in d.mask[n:]) and all(int(x) == 0 for x in d.mask_hi[n:])
        assert list(d.position[:n]) == b["tree_position_ids"].reshape(-1).tolist()
        ret = b["tree_retrieve_indices"]
        assert (d.n_leaves, d.max_depth) == ret.shape
        assert np.asarray(d.retrieve[:d.n_leaves * d.max_depth]).reshape(ret.shape).tolist() == ret.tolist()


def small_llama(layers=2, seed=0):
    """a decoder narrow enough for an fp32 HF twin, with the real head geometry (head_dim 128)"""
    from test_gpu_lm_shapes import hf_llama
    cfg = dict(hidden_size=1024, intermediate_size=2816, num_hidden_layers=layers, num_attention_heads=8, num_key_value_heads=8, vocab_size=4096,
               max_position_embeddings=2048, rms_norm_eps=1e-6)
    return hf_llama(cfg, seed=seed), cfg


@pytest.mark.parametrize("n,shape", [(128, "bushy"), (100, "random"), (65, "chain")])
def test_verify_forward_at_128_rows_matches_hf_fp32(n, shape):
    """LlamaRunner.verify on the 128-row bucket (library GEMMs, two attention tiles) against fp32 HuggingFace with the reference's 4-D tree
    mask (SO/model_patch/llama.py:82-96), and against the SAME runner's 64-row bucket on the first 64 nodes (node i's logits depend on its
    ancestors only, and the first 64 nodes of a tree are a tree)."""
    from transformers import DynamicCache
    from samd_hip.llama import LlamaRunner
    from test_gpu_lm_shapes import tree_mask_4d
    lm, cfg = small_llama()
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)
    rng = np.random.default_rng(n)
    L = 300
    prompt = rng.integers(3, 4096, L).tolist()
    sess = samd_hip.Session(1024)
    ids = torch.tensor([prompt], device="cuda")
    runner.prefill(sess, ids)
    anc = random_parents(rng, n, shape)
    toks = rng.integers(3, 4096, n).tolist()
    sess.set_draft(dev(toks), dev(anc), n, type_=1)
    depth = list(sess.read_draft().position[:n])
    assert runner.bucket(n) == 128
    b = runner.verify(sess, 128)
    torch.cuda.synchronize()
    got = b["logits"][:n].float().clone()
    am = b["argmax"][:n].clone()
    with torch.no_grad():
        cache = DynamicCache()
        lm(input_ids=ids, past_key_values=cache, use_cache=True, logits_to_keep=1)
        want = lm(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[L + x for x in depth]], device="cuda"),
                  attention_mask=tree_mask_4d(anc, L, n), past_key_values=cache, use_cache=True).logits[0]
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    print(f"128-row bucket, n={n} {shape}: |dlogit| {err:.4f} of {scale:.2f}")
    assert err < 0.02 * max(scale, 1.0)
    top2 = want.topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert bool((am.long() == want.argmax(-1))[decided].all()) and int(decided.sum()) > n // 2
    # the first 64 nodes through the 64-row bucket (streaming GEMMs): same logits up to the two GEMM paths' rounding
    sess.set_draft(dev(toks[:64]), dev(anc[:64]), 64, type_=1)
    b64 = runner.verify(sess, 64)
    torch.cuda.synchronize()
    assert (b64["logits"][:64].float() - got[:64]).abs().max().item() < 0.02 * max(scale, 1.0)


def test_generate_with_128_node_drafts_on_a_real_decoder():
    """samd_sam_only at max_predicts 128 / alpha 8 with the FULL verify forward of a real decoder (every kernel, every weight byte) and the
    per-node arg-max of a text that copies long spans of its prompt (samd_hip.engine.ScriptedAcceptance, what bench.py does): drafts of more
    than 64 nodes are verified on the 128-row bucket and accepted beyond 64 tokens in one step.  Output tokens equal the autoregressive run's
    (the reference's criterion, evaluation/equal.py), and so does the KV cache the wide steps leave behind -- rows written by the 128-row
    forward (library GEMMs) and compacted by accept lengths above 64 against rows written one token at a time (streaming GEMMs)."""
    import samd_sam_only as SO
    from samd_hip.engine import ScriptedAcceptance
    from samd_hip.llama import LlamaRunner
    lm, cfg = small_llama(seed=3)
    V, max_len = cfg["vocab_size"], 2048
    rng = np.random.default_rng(11)
    prompt = rng.integers(3, V, 300).tolist()
    cont = []
    while len(cont) < 460:
        q = int(rng.integers(0, 150))
        cont += prompt[q:q + int(rng.integers(70, 140))] + rng.integers(3, V, int(rng.integers(1, 3))).tolist()
    target = prompt + cont
    kv, outs = {}, {}
    for mp in (1, 128):
        runner = LlamaRunner.from_hf(lm, max_cache_len=max_len, dtype=torch.float16, share_weights=False)
        sa = ScriptedAcceptance(runner, V, max_len)
        sa.set_target(target)
        cfg_so = SO.SamdConfig(max_predicts=mp, alpha=8.0, K=8, len_bias=0)
        model = SO.SamdModel(cfg_so, sa, SO.DraftModel(cfg_so, device="cuda"), 2, torch.float16, "cuda")
        gcfg = SO.SamdGenerationConfig(max_new_tokens=400, max_cache_len=max_len)
        model.set_cache(gcfg)
        out = model.generate(torch.tensor([prompt], dtype=torch.long, device="cuda"), generation_config=gcfg)
        torch.cuda.synchronize()
        outs[mp] = (out.output_ids[0], out.accepet_length_per_step, dict(model.engine.bucket_steps))
        kv[mp] = torch.stack(runner.kv_rows(len(prompt) + 380), 1).float().clone()
        del model, sa, runner
        torch.cuda.empty_cache()
    ar, spec = outs[1], outs[128]
    print(f"speculative steps {len(spec[1])}, accept lengths {spec[1]}, buckets {spec[2]}")
    assert ar[0] == spec[0] == target[:len(ar[0])]
    assert spec[2].get(128, 0) >= 2, spec[2]                                  # drafts above 64 nodes were verified ...
    assert max(spec[1]) > 64                                                  # ... and accepted beyond 64 tokens in one step
    err = (kv[1] - kv[128]).abs().max().item()
    assert err < 0.03 * max(1.0, kv[1].abs().max().item()), err              # the cache after wide accepts == the cache written token by token
