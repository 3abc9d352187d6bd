"""The verify forward's glue kernels and the whole forward AT THE BENCHMARKED SHAPES (Vicuna-7B: hidden 4096, inter 11008,
32 heads = 32 KV heads; Llama-3-8B: inter 14336, 8 KV heads, bf16), each through the C ABI against plain PyTorch fp32 arithmetic
of the same op (HF LlamaRMSNorm / apply_rotary_pos_emb / LlamaMLP as driven by samd_sam_only/samd_model.py:134-138 and
samd_sam_only/model_patch/llama.py:82-96).

LM numerics are parity-UNPINNED by the reference (the arithmetic lives in HuggingFace transformers, no fixtures);
these tests pin them against transformers 5.x in fp32 on the same GPU.  Tolerances (stated per test): one rounding of the
model dtype on values of magnitude ~1 -- fp16 2^-11 ~ 5e-4 relative, bf16 2^-8 ~ 4e-3 relative -- times a small factor for the
rsqrt / exp approximations."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip import _ptr, check, current_stream, lib, torch_dtype_code
from util import random_parents

RTOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def rel_err(got, want):
    return ((got.float() - want.float()).abs().max() / want.float().abs().max().clamp_min(1e-6)).item()


def gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,n_part", [(1, 0), (16, 0), (64, 0), (16, 8), (64, 2), (16, -1)])
def test_rmsnorm_hidden_4096(dtype, rows, n_part):
    """samd_rmsnorm at hidden 4096 (the 512-thread path): residual add from a T tensor, from fp32 split-K partial sums (n_part
    splits, as samd_gemm_skinny leaves them), or no residual (n_part = -1).  x_out must be bit-exact; the norm output within
    RTOL of w * (x * rsqrt(mean(x^2) + eps)).to(dtype) computed in fp32."""
    H, eps, g = 4096, 1e-6, gen(rows * 10 + n_part + 3)
    x = (torch.randn((rows, H), generator=g, device="cuda") * 3).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g, device="cuda")).to(dtype)
    out = torch.empty_like(x)
    x_in = x.clone()
    if n_part > 0:
        part = torch.randn((n_part, rows, H), generator=g, device="cuda", dtype=torch.float32)
        delta_t = part.sum(0).to(dtype)                                  # the sum is rounded to T like the GEMM's own epilogue
        check(lib().samd_rmsnorm(_ptr(x), _ptr(part), _ptr(w), _ptr(out), rows, H, eps, torch_dtype_code(dtype), n_part, rows * H, current_stream()))
    elif n_part == 0:
        delta_t = torch.randn((rows, H), generator=g, device="cuda").to(dtype)
        check(lib().samd_rmsnorm(_ptr(x), _ptr(delta_t), _ptr(w), _ptr(out), rows, H, eps, torch_dtype_code(dtype), 0, 0, current_stream()))
    else:
        delta_t = None
        check(lib().samd_rmsnorm(_ptr(x), None, _ptr(w), _ptr(out), rows, H, eps, torch_dtype_code(dtype), 0, 0, current_stream()))
    torch.cuda.synchronize()
    x_want = x_in if delta_t is None else (x_in.float() + delta_t.float()).to(dtype)
    if n_part > 0:
        # fp32 summation order differs from torch's; the rounded sum may differ by one ulp of T on a few elements
        assert rel_err(x, x_want) < RTOL[dtype]
        x_want = x.clone()
    else:
        assert torch.equal(x, x_want)                                    # HF: hidden_states = residual + hidden_states
    xf = x_want.float()
    normed = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)
    want = (w.float() * normed.float()).to(dtype)
    assert rel_err(out, want) < RTOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,rows,n,n_part", [(32, 32, 16, 11, 0), (32, 32, 64, 60, 2), (32, 8, 16, 16, 0), (32, 8, 64, 37, 2), (32, 8, 1, 1, 0)])
def test_rope_kv_write_real_heads(dtype, H, Hkv, rows, n, n_part):
    """samd_rope_kv_write at 32 query heads / 32 and 8 KV heads, head_dim 128: q_out = rope(q), K cache rows [L, L+n) = rope(k),
    V rows = v (SamdStaticCache.update, cache.py:103-115), positions L + tree depth; rows >= n and every other cache row untouched.
    Reference: HF apply_rotary_pos_emb (rotate_half) in fp32 from the same fp32 cos/sin tables, one rounding to dtype."""
    D, max_len, L, g = 128, 2048, 777, gen(H + Hkv + rows + n_part)
    W = (H + 2 * Hkv) * D
    if n_part:
        part = torch.randn((n_part, rows, W), generator=g, device="cuda", dtype=torch.float32)
        qkv_t, src = part.sum(0).to(dtype), part
    else:
        qkv_t = torch.randn((rows, W), generator=g, device="cuda").to(dtype)
        src = qkv_t
    rel = torch.tensor(np.random.default_rng(1).integers(0, 9, 64), dtype=torch.int32, device="cuda")
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    ang = torch.outer(torch.arange(max_len, dtype=torch.float64), inv)
    cos, sin = ang.cos().float().cuda().contiguous(), ang.sin().float().cuda().contiguous()
    q_out = torch.full((rows, H, D), 5.0, device="cuda").to(dtype)
    kc = torch.full((Hkv, max_len, D), 7.0, device="cuda").to(dtype)
    vc = torch.full((Hkv, max_len, D), 9.0, device="cuda").to(dtype)
    d_L, d_n = torch.tensor([L], dtype=torch.int32, device="cuda"), torch.tensor([n], dtype=torch.int32, device="cuda")
    check(lib().samd_rope_kv_write(_ptr(src), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cos), _ptr(sin), _ptr(q_out), _ptr(kc), _ptr(vc), rows, H, Hkv, D,
                                   max_len, max_len, torch_dtype_code(dtype), n_part, rows * W, current_stream()))
    torch.cuda.synchronize()
    x = qkv_t[:n].float().view(n, H + 2 * Hkv, D)
    pos = (L + rel[:n].long())
    c, s = torch.cat((cos[pos], cos[pos]), -1)[:, None, :], torch.cat((sin[pos], sin[pos]), -1)[:, None, :]
    rot = torch.cat((-x[..., D // 2:], x[..., :D // 2]), -1)
    roped = (x * c + rot * s)
    tol = RTOL[dtype] * (2 if n_part else 1)
    assert rel_err(q_out[:n], roped[:, :H].to(dtype)) < tol
    assert rel_err(kc[:, L:L + n], roped[:, H:H + Hkv].to(dtype).transpose(0, 1)) < tol
    if n_part:
        assert rel_err(vc[:, L:L + n], x[:, H + Hkv:].to(dtype).transpose(0, 1)) < tol
    else:
        assert torch.equal(vc[:, L:L + n], qkv_t[:n].view(n, H + 2 * Hkv, D)[:, H + Hkv:].transpose(0, 1))
    assert (kc[:, :L] == 7).all() and (kc[:, L + n:] == 7).all() and (vc[:, :L] == 9).all() and (vc[:, L + n:] == 9).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("vocab", [32000, 128256])
def test_embed_rows_real_vocab(dtype, vocab):
    """samd_embed_rows: out[r] = table[tokens[r]] bit-exact at hidden 4096 (ids outside the table are clamped, never read out of bounds)."""
    H, g = 4096, gen(vocab)
    table = torch.randn((vocab, H), generator=g, device="cuda").to(dtype)
    toks = torch.tensor([0, 1, 2, vocab - 1, vocab // 2, 31999, 5, 5] + list(range(100, 156)), dtype=torch.int32, device="cuda")
    out = torch.zeros((64, H), dtype=dtype, device="cuda")
    check(lib().samd_embed_rows(_ptr(toks), _ptr(table), _ptr(out), 64, H, vocab, torch_dtype_code(dtype), current_stream()))
    assert torch.equal(out, table[toks.long()])
    bad = torch.tensor([-3, vocab + 7] + [1] * 14, dtype=torch.int32, device="cuda")
    check(lib().samd_embed_rows(_ptr(bad), _ptr(table), _ptr(out), 16, H, vocab, torch_dtype_code(dtype), current_stream()))
    assert torch.equal(out[0], table[0]) and torch.equal(out[1], table[vocab - 1])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("inter,rows,n_part", [(11008, 16, 0), (11008, 64, 4), (14336, 16, 0), (14336, 32, 2)])
def test_silu_mul_real_inter(dtype, inter, rows, n_part):
    """samd_silu_mul: act_fn(gate) * up with HF's roundings (silu rounded to dtype, then the product), inter 11008 / 14336."""
    g = gen(inter + rows)
    if n_part:
        part = torch.randn((n_part, rows, 2 * inter), generator=g, device="cuda", dtype=torch.float32)
        gu_t, src = part.sum(0).to(dtype), part
    else:
        gu_t = (torch.randn((rows, 2 * inter), generator=g, device="cuda") * 2).to(dtype)
        src = gu_t
    out = torch.empty((rows, inter), dtype=dtype, device="cuda")
    check(lib().samd_silu_mul(_ptr(src), _ptr(out), rows, inter, torch_dtype_code(dtype), n_part, rows * 2 * inter, current_stream()))
    gate, up = gu_t[:, :inter], gu_t[:, inter:]
    want = (torch.nn.functional.silu(gate.float()).to(dtype).float() * up.float()).to(dtype)
    assert rel_err(out, want) < RTOL[dtype] * (2 if n_part else 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("inter,rows", [(11008, 16), (14336, 64)])
def test_gemm_silu_epilogue_real_inter(dtype, inter, rows):
    """samd_gemm_skinny_silu (gate|up projection + SiLU*up epilogue) == LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) at hidden 4096.
    Tolerance: K = 4096 fp32-accumulated products rounded once to dtype, then the activation's two roundings."""
    Hd, g = 4096, gen(inter)
    x = torch.randn((rows, Hd), generator=g, device="cuda").to(dtype)
    wg = (torch.randn((inter, Hd), generator=g, device="cuda") * 0.02).to(dtype)
    wu = (torch.randn((inter, Hd), generator=g, device="cuda") * 0.02).to(dtype)
    inter_leaved = torch.stack([wg.view(inter // 64, 64, Hd), wu.view(inter // 64, 64, Hd)], dim=1).reshape(2 * inter, Hd).contiguous()
    packed = torch.empty_like(inter_leaved)
    check(lib().samd_gemm_pack_weights(_ptr(inter_leaved), _ptr(packed), 2 * inter, Hd, current_stream()))
    out = torch.empty((rows, inter), dtype=dtype, device="cuda")
    check(lib().samd_gemm_skinny_silu(_ptr(x), _ptr(packed), rows, 2 * inter, Hd, _ptr(out), torch_dtype_code(dtype), current_stream()))
    gate, up = (x.float() @ wg.float().t()).to(dtype), (x.float() @ wu.float().t()).to(dtype)
    want = (torch.nn.functional.silu(gate.float()).to(dtype).float() * up.float()).to(dtype)
    assert rel_err(out, want) < 3 * RTOL[dtype]


# ---------------------------------------------------------------------------------------------------------------------
def hf_llama(cfg_kw, seed, std=0.02):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(**cfg_kw, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    with torch.device("cuda"):
        lm = LlamaForCausalLM(cfg)
    lm = lm.float().eval()
    g = gen(seed)
    with torch.no_grad():
        for name, p in lm.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") * std)
            else:
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=g, device="cuda"))
    return lm


def tree_mask_4d(anc, L, n):
    mask = torch.full((1, 1, n, L + n), torch.finfo(torch.float32).min, device="cuda")
    mask[..., :L] = 0
    for i in range(n):
        j = i
        while j != -1:
            mask[0, 0, i, L + j] = 0
            j = anc[j]
    return mask


def verify_against_hf(lm, runner, prompt_len, n, vocab, tol, seed=1, decide_gap=None):
    from transformers import DynamicCache
    rng = np.random.default_rng(seed)
    sess = samd_hip.Session(prompt_len + 128)
    prompt = rng.integers(3, vocab, prompt_len).tolist()
    ids = torch.tensor([prompt], device="cuda")
    last = runner.prefill(sess, ids)
    with torch.no_grad():
        cache = DynamicCache()
        ref_last = lm(input_ids=ids, past_key_values=cache, use_cache=True, logits_to_keep=1).logits[0, -1]
    err_prefill = (last.float() - ref_last).abs().max().item()
    anc = random_parents(rng, n, "bushy")
    toks = rng.integers(3, vocab, n).tolist()
    dev = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int32)).cuda()
    sess.set_draft(dev(toks), dev(anc), n, type_=1)
    depth = list(sess.read_draft().position[:n])
    b = runner.verify(sess, runner.bucket(n))
    torch.cuda.synchronize()
    got = b["logits"][:n].float()
    L = prompt_len
    with torch.no_grad():
        want = lm(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[L + x for x in depth]], device="cuda"),
                  attention_mask=tree_mask_4d(anc, L, n), past_key_values=cache, use_cache=True).logits[0]
    err_tree = (got - want).abs().max().item()
    top2 = want.topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > (decide_gap or 4 * (tol or 0.0))      # rows whose fp32 arg-max is not a near-tie
    agree = (b["argmax"][:n].long() == want.argmax(-1))
    print(f"prefill-last |dlogit| {err_prefill:.4f}, tree |dlogit| {err_tree:.4f} (|logit| max {want.abs().max().item():.2f}); arg-max agreement "
          f"{agree.float().mean().item():.3f} over {n} nodes, {int(decided.sum())} decided rows")
    if tol is not None:
        assert err_prefill < tol and err_tree < tol
        assert bool(agree[decided].all())
    verify_against_hf.last = dict(argmax=b["argmax"][:n].long().clone(), want=want, ref_last=ref_last.float(), prompt=prompt, toks=toks, anc=anc, depth=depth)
    return err_prefill, err_tree


def hf_low_precision_twin(lm, dtype):
    """the reference's own arithmetic: the same HF module cast to the serving dtype (what SO/samd_model.py:134-138 runs), rotary
    frequencies kept in fp32 as from_pretrained(dtype=...) keeps them"""
    import copy
    lm16 = copy.deepcopy(lm).to(dtype)
    lm16.model.rotary_emb.inv_freq = lm.model.rotary_emb.inv_freq.clone()
    if hasattr(lm16.model.rotary_emb, "original_inv_freq"):
        lm16.model.rotary_emb.original_inv_freq = lm.model.rotary_emb.inv_freq.clone()
    return lm16


def check_against_reference_arithmetic(lm, lm16, runner, prompt_len, n, vocab, seed, label):
    """OUR forward against fp32 HuggingFace, with the yardstick the reference itself sets: the same model run by HF in the serving dtype.
    Asserts err(ours, fp32) <= 1.5 x err(HF-low, fp32) (+ 0.02) after the prefill and on the tree rows, and arg-max agreement with
    HF-LOW-PRECISION -- the tokens the reference would emit -- on every node whose fp32 top-2 gap exceeds twice that error."""
    from transformers import DynamicCache
    e_pre, e_tree = verify_against_hf(lm, runner, prompt_len, n, vocab, tol=None, seed=seed)
    c = verify_against_hf.last
    ids = torch.tensor([c["prompt"]], device="cuda")
    with torch.no_grad():
        cache = DynamicCache()
        last16 = lm16(input_ids=ids, past_key_values=cache, use_cache=True, logits_to_keep=1).logits[0, -1].float()
        mask = tree_mask_4d(c["anc"], prompt_len, n).to(next(lm16.parameters()).dtype)
        tree16 = lm16(input_ids=torch.tensor([c["toks"]], device="cuda"), position_ids=torch.tensor([[prompt_len + x for x in c["depth"]]], device="cuda"),
                      attention_mask=mask, past_key_values=cache, use_cache=True).logits[0].float()
    hf_pre = (c["ref_last"] - last16).abs().max().item()
    hf_tree = (c["want"] - tree16).abs().max().item()
    print(f"{label}, L={prompt_len}, n={n}: ours vs fp32 {e_pre:.4f} / {e_tree:.4f}; HF low-precision vs fp32 {hf_pre:.4f} / {hf_tree:.4f}")
    assert e_pre <= 1.5 * hf_pre + 0.02 and e_tree <= 1.5 * hf_tree + 0.02, (e_pre, hf_pre, e_tree, hf_tree)
    top2 = c["want"].topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * max(e_tree, hf_tree)
    agree16 = c["argmax"] == tree16.argmax(-1)
    print(f"    arg-max == HF low-precision on {int(agree16.sum())}/{n} nodes; {int(decided.sum())} nodes have an fp32 top-2 gap > {2 * max(e_tree, hf_tree):.3f}")
    assert int(decided.sum()) >= max(1, n // 8)                                # the criterion must bite (random-init logits are flat: ~1/4 of the nodes)
    assert bool(agree16[decided].all())
    assert int(agree16.sum()) >= int(0.85 * n)                                 # and near-ties aside, we emit the reference's tokens
    return e_pre, e_tree, hf_pre, hf_tree


def test_vicuna_7b_shape_forward_matches_hf_fp32():
    """The forward bench.py times -- 32 layers, hidden 4096, 32 heads, inter 11008, vocab 32000, fp16 -- against a random-init
    transformers LlamaForCausalLM of the same shape in fp32 (27 GB) on the same GPU: last-position logits after a 1000-token
    prefill, then a 60-node tree verify over the cached prompt with the reference's 4-D additive tree mask
    (samd_sam_only/model_patch/llama.py:82-96), then a 13-node tree on the 16-row bucket (the norm-fold forward).  Round 5:
    no absolute tolerance any more -- the yardstick is the reference's OWN arithmetic, the same HF model cast to fp16 (what
    SO/samd_model.py:134-138 runs on Vicuna): our error against fp32 must stay within 1.5x of HF-fp16's error against fp32, and our
    arg-max must equal HF-fp16's on every node whose fp32 top-2 gap exceeds twice that error (check_against_reference_arithmetic; the
    bf16 Llama-3 test below uses the same criterion).  Measured on MI355X: ours 0.070 / 0.081 on logits of magnitude ~6.5."""
    from samd_hip.llama import LlamaRunner
    cfg = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32,
               vocab_size=32000, max_position_embeddings=2048, rms_norm_eps=1e-6)
    lm = hf_llama(cfg, seed=0)
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)
    assert runner.norm_fold
    lm16 = hf_low_precision_twin(lm, torch.float16)
    check_against_reference_arithmetic(lm, lm16, runner, 1000, 60, 32000, seed=1, label="vicuna-7b fp16, 64-row bucket")
    check_against_reference_arithmetic(lm, lm16, runner, 700, 13, 32000, seed=2, label="vicuna-7b fp16, 16-row bucket (norm-fold)")


def test_norm_fold_forward_equals_the_eight_launch_forward(monkeypatch):
    """the 16-row forward in its two forms on the same weights (Vicuna-7B width, 4 layers): six launches per layer (norm-fold) against
    eight (k_rmsnorm + split-K partials).  Same roundings of the same quantities, so the logits differ only by the summation order of
    the projections and of the rows' sums of squares: a few fp16 ulps of the logit scale; the arg-max agrees wherever the top-2 gap
    exceeds that."""
    from samd_hip.llama import LlamaRunner
    cfg = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=32,
               vocab_size=32000, max_position_embeddings=2048, rms_norm_eps=1e-6)
    rng = np.random.default_rng(4)
    prompt = torch.tensor([rng.integers(3, 32000, 300).tolist()], device="cuda")
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SAMD_NORM_FOLD", mode)
        runner = LlamaRunner.random_init(cfg, 2048, torch.float16, seed=5)
        assert runner.norm_fold == (mode == "1")
        sess = samd_hip.Session(600)
        runner.prefill(sess, prompt)
        res = []
        for n in (1, 7, 16):
            anc = random_parents(np.random.default_rng(n), n, "bushy")
            toks = np.random.default_rng(100 + n).integers(3, 32000, n).tolist()
            dev = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int32)).cuda()
            sess.set_draft(dev(toks), dev(anc), n, type_=1)
            b = runner.verify(sess, runner.bucket(n))
            torch.cuda.synchronize()
            res.append((b["logits"][:n].float().clone(), b["argmax"][:n].clone(), b["x"][:n].float().clone()))
        outs[mode] = res
        del runner
        torch.cuda.empty_cache()
    for (la, aa, xa), (lb, ab, xb) in zip(outs["0"], outs["1"]):
        scale = max(1.0, la.abs().max().item())
        assert (la - lb).abs().max().item() <= 0.02 * scale
        assert (xa - xb).abs().max().item() <= 0.02 * max(1.0, xa.abs().max().item())
        top2 = la.topk(2, dim=-1).values
        decided = (top2[:, 0] - top2[:, 1]) > 0.04 * scale
        assert bool((aa == ab)[decided].all())


def test_llama3_shape_long_context_matches_hf_fp32():
    """Llama-3-8B's head geometry (32 query / 8 KV heads, head_dim 128, llama3 rope scaling, 8192 positions, bf16) at L ~ 6000:
    4 layers of the real width (hidden 4096, inter 14336) so that the fp32 HF twin's eager 6000 x 6000 attention fits;
    6000-token prefill (one-pass path), then a 63-node tree verify at L = 6000.  bf16 tolerance 0.25 absolute on logits."""
    from samd_hip.llama import LlamaRunner
    rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192,
              rope_theta=500000.0)
    cfg = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=8,
               vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_parameters=rs)
    try:
        lm = hf_llama(cfg, seed=2)
    except Exception as e:
        pytest.skip(f"LlamaConfig(rope_parameters=...) unsupported: {e}")
    for attention in ("split", "block"):
        runner = LlamaRunner.from_hf(lm, max_cache_len=8192, dtype=torch.bfloat16, attention=attention)
        verify_against_hf(lm, runner, 6000, 63, 128256, tol=0.25, seed=3)
        del runner


def test_llama3_8b_full_depth_forward_matches_hf_fp32():
    """configs[3]'s base model at FULL depth: Llama-3-8B's shape -- 32 layers, hidden 4096, inter 14336, 32 query / 8 KV heads,
    vocab 128256, llama3 rope scaling, bf16 -- against a random-init transformers LlamaForCausalLM of the same shape in fp32
    (32 GB) on the same GPU: last-position logits after a 2000-token prefill, then a 63-node tree verify (EAGLE-2's draft size) over
    the cached prompt with the reference's 4-D additive tree mask (samd_sam_only/model_patch/llama.py:82-96), then a 13-node tree on
    the 16-row bucket (norm-fold forward).  bf16 rounds activations to 8 bits ~250 times on the way through 32 layers, so the yardstick
    is what the REFERENCE's own arithmetic loses in bf16: the same HF model cast to bfloat16 (what samd/samd_model.py runs) is
    evaluated beside ours, and our error against fp32 must stay within 1.5x of HF-bf16's error against fp32 (plus 0.02); the absolute
    bound TOL = 0.8 on logits of range +-6.7 is stated for the record (measured on MI355X: 0.39 after the prefill, 0.67 on the tree
    rows).  The arg-max must agree with fp32 on every node whose fp32 top-2 gap exceeds 2 x TOL."""
    import copy
    from transformers import DynamicCache
    from samd_hip.llama import LlamaRunner
    TOL = 0.8
    rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192,
              rope_theta=500000.0)
    cfg = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
               vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_parameters=rs)
    try:
        lm = hf_llama(cfg, seed=3)
    except Exception as e:
        pytest.skip(f"LlamaConfig(rope_parameters=...) unsupported: {e}")
    runner = LlamaRunner.from_hf(lm, max_cache_len=4096, dtype=torch.bfloat16, share_weights=False)
    assert runner.norm_fold
    # the reference's own arithmetic in bf16: HF's LlamaForCausalLM cast to bfloat16, same prompt, same tree
    lm16 = copy.deepcopy(lm).to(torch.bfloat16)
    lm16.model.rotary_emb.inv_freq = lm.model.rotary_emb.inv_freq.clone()       # .to(bf16) rounds the rotary frequencies too; from_pretrained(dtype=bf16) keeps them fp32
    if hasattr(lm16.model.rotary_emb, "original_inv_freq"):
        lm16.model.rotary_emb.original_inv_freq = lm.model.rotary_emb.inv_freq.clone()
    for prompt_len, n, seed in ((2000, 63, 5), (1200, 13, 6)):
        e_pre, e_tree = verify_against_hf(lm, runner, prompt_len, n, 128256, tol=TOL, seed=seed, decide_gap=2 * TOL)
        rng = np.random.default_rng(seed)
        prompt = rng.integers(3, 128256, prompt_len).tolist()
        anc = random_parents(rng, n, "bushy")
        toks = rng.integers(3, 128256, n).tolist()
        depth = [0] * n
        for i in range(1, n):
            depth[i] = depth[anc[i]] + 1
        ids = torch.tensor([prompt], device="cuda")
        outs = []
        with torch.no_grad():
            for m in (lm, lm16):
                cache = DynamicCache()
                last = m(input_ids=ids, past_key_values=cache, use_cache=True, logits_to_keep=1).logits[0, -1].float()
                mask = tree_mask_4d(anc, prompt_len, n).to(next(m.parameters()).dtype)
                tree = m(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[prompt_len + x for x in depth]], device="cuda"),
                         attention_mask=mask, past_key_values=cache, use_cache=True).logits[0].float()
                outs.append((last, tree))
        hf_pre = (outs[0][0] - outs[1][0]).abs().max().item()
        hf_tree = (outs[0][1] - outs[1][1]).abs().max().item()
        print(f"llama3-8b full depth, L={prompt_len}, n={n}: ours vs fp32 {e_pre:.4f} / {e_tree:.4f}; HF bf16 vs fp32 {hf_pre:.4f} / {hf_tree:.4f}")
        assert e_pre <= 1.5 * hf_pre + 0.02 and e_tree <= 1.5 * hf_tree + 0.02


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H,Hkv,L,n", [(32, 8, 6000, 63), (32, 32, 8192 - 64, 64), (32, 8, 4097, 17)])
def test_tree_attention_long_context(dtype, tol, H, Hkv, L, n):
    """samd_tree_attention beyond 2048 positions (Llama-3-8B's 8192): fp32 SDPA reference with the tree mask."""
    from test_gpu_verify import reference_attention
    from oracle import sam_oracle as O
    rng = np.random.default_rng(L + n)
    D, max_len, n_pad = 128, 8192, 64
    g = gen(L + n)
    q = torch.randn((n_pad, H, D), generator=g, device="cuda").to(dtype)
    kc = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    vc = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    kc[:, L + n:] = float("nan")
    vc[:, L + n:] = float("nan")
    anc = random_parents(rng, n, "bushy")
    m = O.gen_buffers(anc)["tree_attn_mask"][0, 0]
    rows = [int(sum(1 << j for j in range(n) if m[i, j])) for i in range(n)]
    mask = torch.tensor(np.array(rows + [0] * (64 - n), dtype=np.uint64).view(np.int64), device="cuda")
    out = torch.zeros((n_pad, H, D), dtype=dtype, device="cuda")
    ws_bytes = lib().samd_tree_attention_workspace(n_pad, H, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    d_L, d_n = torch.tensor([L], dtype=torch.int32, device="cuda"), torch.tensor([n], dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(D)
    check(lib().samd_tree_attention(_ptr(q), _ptr(kc), _ptr(vc), _ptr(out), torch_dtype_code(dtype), n_pad, H, Hkv, D, max_len, _ptr(mask), _ptr(d_L),
                                    _ptr(d_n), scale, _ptr(ws), ws_bytes, current_stream()))
    want = reference_attention(q, kc, vc, L, n, rows, scale)
    err = (out[:n].float() - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err
