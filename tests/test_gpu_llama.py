"""GPU parity of the verify forward (samd_hip.llama.LlamaRunner) against HuggingFace's LlamaForCausalLM in fp32 with
an explicit 4-D additive tree mask (the semantics of samd_sam_only/model_patch/llama.py:82-96), and the losslessness
property the reference checks with evaluation/equal.py: speculative output == autoregressive greedy output of the same
kernels (up to fp16 arg-max near-ties, which the test identifies and bounds).

Tolerance: the runner computes in fp16 with fp32 accumulation; logits of this 2-layer model agree with the fp32
reference to 3e-2 absolute (|logit| ~ 1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

import samd_hip
from samd_hip.llama import LlamaRunner
from util import random_parents

TOL = 3e-2


def tiny_llama(kv_heads=2, seed=0, vocab=512):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=kv_heads,
                      vocab_size=vocab, max_position_embeddings=512, head_dim=128, rms_norm_eps=1e-5, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    lm = LlamaForCausalLM(cfg).cuda().float().eval()
    return lm


def dev(a):
    return torch.as_tensor(np.asarray(a, dtype=np.int32)).cuda()


@pytest.mark.parametrize("attention", ["split", "split2", "split3", "block"])
@pytest.mark.parametrize("kv_heads", [2, 1])
def test_prefill_and_tree_verify_match_hf(kv_heads, attention):
    """both attention modes of the runner: "split" (RoPE / split attention / merge launches; V cached transposed since round 6, "split2" keeps it row-major) and "block"
    (samd_attention_block: one launch, V cached transposed)"""
    from transformers import DynamicCache
    lm = tiny_llama(kv_heads)
    runner = LlamaRunner.from_hf(lm, max_cache_len=256, dtype=torch.float16, attention=attention)
    sess = samd_hip.Session(512)
    rng = np.random.default_rng(1)
    prompt = rng.integers(3, 512, 75).tolist()
    ids = torch.tensor([prompt], device="cuda")
    last = runner.prefill(sess, ids)
    with torch.no_grad():
        cache = DynamicCache()
        ref = lm(input_ids=ids, past_key_values=cache, use_cache=True).logits[0]
    assert (last.float() - ref[-1]).abs().max().item() < TOL
    assert sess.get_cache_length() == len(prompt)
    # tree verify of a random 23-node tree over the cached prompt
    n, L = 23, len(prompt)
    anc = random_parents(rng, n, "bushy")
    toks = rng.integers(3, 512, n).tolist()
    sess.set_draft(dev(toks), dev(anc), n, type_=1)
    d = sess.read_draft()
    b = runner.verify(sess, runner.bucket(n))
    torch.cuda.synchronize()
    got = b["logits"][:n].float()
    depth = list(d.position[:n])
    mask = torch.full((1, 1, n, L + n), torch.finfo(torch.float32).min, device="cuda")
    mask[..., :L] = 0
    for i in range(n):
        j = i
        while j != -1:
            mask[0, 0, i, L + j] = 0
            j = anc[j]
    with torch.no_grad():
        want = lm(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[L + x for x in depth]], device="cuda"),
                  attention_mask=mask, past_key_values=cache, use_cache=True).logits[0]
    assert (got - want).abs().max().item() < TOL
    assert (b["argmax"][:n].cpu() == want.argmax(-1).cpu()).float().mean().item() > 0.8


def test_projection_that_cannot_stream_falls_back_alone():
    """a 515-row lm_head (not a multiple of 128) goes to the library GEMM; the layer projections keep the streaming kernel"""
    lm = tiny_llama(2, vocab=515)
    runner = LlamaRunner.from_hf(lm, max_cache_len=256, dtype=torch.float16)
    assert runner.wp is not None and runner.wp["lm_head"] is None and runner.wp["layers"][0]["wqkv"] is not None and runner.fused_mlp
    sess = samd_hip.Session(512)
    prompt = np.random.default_rng(3).integers(3, 515, 40).tolist()
    ids = torch.tensor([prompt], device="cuda")
    last = runner.prefill(sess, ids)
    with torch.no_grad():
        ref = lm(input_ids=ids).logits[0]
    assert (last.float() - ref[-1]).abs().max().item() < TOL


def _near_tie(lm, prefix, a, b, eps=5e-2):
    with torch.no_grad():
        lg = lm(input_ids=torch.tensor([prefix], device="cuda")).logits[0, -1]
    return abs(lg[a].item() - lg[b].item()) < eps


def test_speculative_equals_autoregressive():
    """evaluation/equal.py's criterion on a seeded tiny Llama: SAM-drafted decoding returns the AR greedy sequence."""
    import samd_sam_only as SO
    lm = tiny_llama(2, seed=3)
    rng = np.random.default_rng(2)
    prompt = rng.integers(3, 512, 40).tolist()
    ids = torch.tensor([prompt], device="cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=96, max_cache_len=512)
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, lm, SO.DraftModel(ar_cfg, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
    out_ar = ar.generate(ids, generation_config=gcfg)
    assert out_ar.decode_steps == out_ar.decode_tokens                       # one token per step
    seq_ar = out_ar.output_ids[0]
    # a corpus that contains the continuation (plus every token as a one-token document, as the reference's tool does)
    docs = [seq_ar[len(prompt):]] + [rng.integers(3, 512, 50).tolist() for _ in range(4)] + [[i] for i in range(512)]
    cfg = SO.SamdConfig(max_predicts=16, alpha=4.0, len_bias=0)
    draft = SO.DraftModel(cfg, sam_static=SO.build_sam(docs, 2), device="cuda")
    spec = SO.SamdModel(cfg, lm, draft, eos_token_id=2, dtype=torch.float16, device="cuda")
    for use_graphs in (True, False):
        spec.set_cache(gcfg)
        spec.engine.use_graphs = use_graphs
        out = spec.generate(ids, generation_config=gcfg)
        seq = out.output_ids[0]
        assert out.decode_steps < out.decode_tokens, "drafts were never accepted"
        m = min(len(seq), len(seq_ar))
        diff = [i for i in range(m) if seq[i] != seq_ar[i]]
        if diff:                                                           # only an fp16 near-tie may split the two runs
            i = diff[0]
            assert i > len(prompt) + 8 and _near_tie(lm, seq[:i], seq[i], seq_ar[i]), f"diverged at {i}"
    # the granular form (prefill / decode / update_state) walks the same path
    spec.gen_config = gcfg
    got = list(prompt)
    for new_ids, _ in spec._run_granular(ids, gcfg, 24):
        got.extend(new_ids)
    m = min(len(got), len(seq_ar))
    diff = [i for i in range(m) if got[i] != seq_ar[i]]
    assert not diff or _near_tie(lm, got[:diff[0]], got[diff[0]], seq_ar[diff[0]])


def test_llama3_rope_scaling_matches_hf():
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    from samd_hip.llama import LlamaShape
    rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192,
              rope_theta=500000.0)
    cfg = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
               vocab_size=128256, max_position_embeddings=8192, rope_parameters=rs)
    mine = LlamaShape(cfg).inv_freq()
    try:
        hf_cfg = LlamaConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, max_position_embeddings=8192,
                             rope_parameters=rs)
        want, _ = ROPE_INIT_FUNCTIONS["llama3"](hf_cfg, "cpu")
    except Exception as e:                                                   # HF signature drift: not our arithmetic
        pytest.skip(f"HF rope init unavailable: {e}")
    assert torch.allclose(mine.float(), want.float(), rtol=1e-6, atol=0)


def test_eagle2_plugin_is_lossless_and_paths_agree():
    """samd[EAGLE2] (BASELINE config 4 shape of the loop): SAM sequence drafts when the match is long, otherwise the
    EAGLE-2 head's dynamic 63-node tree.  With a random head the drafts are poor, but verification makes the output the
    autoregressive greedy sequence whatever the draft -- the reference's losslessness criterion (evaluation/equal.py)."""
    import samd as S
    import samd_sam_only as SO
    from samd.tree_model.eagle2 import Eagle2, Eagle2Head
    lm = tiny_llama(2, seed=5)
    rng = np.random.default_rng(4)
    prompt = rng.integers(3, 512, 70).tolist()              # > 64: two prefill chunks feed the head
    prompt[40:52] = prompt[10:22]                           # an in-prompt repeat -> dynamic-SAM sequence drafts
    ids = torch.tensor([prompt], device="cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=48, max_cache_len=512)
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, lm, SO.DraftModel(ar_cfg, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
    seq_ar = ar.generate(ids, generation_config=gcfg).output_ids[0]

    tree_cfg = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, vocab_size=512,
                    rms_norm_eps=1e-5, bias=True)
    cfg = S.SamdConfig(n_predicts=12, len_threshold=3, len_bias=0, tree_method="eagle2", tree_config=tree_cfg)
    assert cfg.use_last_hidden_states

    def build():
        head = Eagle2Head(tree_cfg, dtype=torch.float16, device="cuda")
        head.random_init(seed=3, std=0.08)
        tm = Eagle2(cfg, lm, torch.float16, "cuda", head=head)
        draft = S.DraftModel(cfg, tree_model=tm, lm=lm, device="cuda")
        return S.SamdModel(cfg, lm, draft, eos_token_id=2, dtype=torch.float16, device="cuda")

    def check(seq):
        m = min(len(seq), len(seq_ar))
        diff = [i for i in range(m) if seq[i] != seq_ar[i]]
        assert not diff or (diff[0] > len(prompt) + 4 and _near_tie(lm, seq[:diff[0]], seq[diff[0]], seq_ar[diff[0]])), diff[:3]

    model = build()
    out = model.generate(ids, generation_config=gcfg)
    check(out.output_ids[0])
    assert model.lookup_stats["tree"][0] > 0, "the EAGLE-2 tree path never ran"
    # granular form: prefill / decode / update_state with the plugin's host-visible gen_draft
    model2 = build()
    model2.gen_config = gcfg
    got = list(prompt)
    for new_ids, _ in model2._run_granular(ids, gcfg, 16):
        got.extend(new_ids)
    check(got)
    # a draft straight from the plugin: 63 nodes, parents precede children
    model3 = build()
    model3.set_cache(gcfg)
    rep = model3.engine.start(ids)
    d = model3.draft.session().read_draft()
    if rep.type == 1 and d.n == 63:
        par = list(d.parent[:63])
        assert par[0] == -1 and all(0 <= par[i] < i for i in range(1, 63))


def test_eagle_static_tree_plugin_is_lossless_and_paths_agree():
    """samd[EAGLE] (v1, static 26-node tree): same losslessness criterion as EAGLE-2, fused engine and granular form; the
    installed draft is the static tree (parent array of samd.tree_model.eagle.StaticDraftTree)."""
    import samd as S
    import samd_sam_only as SO
    from samd.tree_model.eagle import Eagle, EagleHead, StaticDraftTree
    lm = tiny_llama(2, seed=6)
    rng = np.random.default_rng(5)
    prompt = rng.integers(3, 512, 70).tolist()
    prompt[40:52] = prompt[10:22]
    ids = torch.tensor([prompt], device="cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=48, max_cache_len=512)
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, lm, SO.DraftModel(ar_cfg, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
    seq_ar = ar.generate(ids, generation_config=gcfg).output_ids[0]
    tree_cfg = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, vocab_size=512,
                    rms_norm_eps=1e-5, bias=True)
    cfg = S.SamdConfig(n_predicts=12, len_threshold=3, len_bias=0, tree_method="eagle", tree_config=tree_cfg)
    assert cfg.use_last_hidden_states and len(cfg.tree) == 25

    def build():
        head = EagleHead(tree_cfg, dtype=torch.float16, device="cuda")
        head.random_init(seed=3, std=0.08)
        head.set_tree(StaticDraftTree(cfg.tree))
        tm = Eagle(cfg, lm, torch.float16, "cuda", head=head)
        draft = S.DraftModel(cfg, tree_model=tm, lm=lm, device="cuda")
        return S.SamdModel(cfg, lm, draft, eos_token_id=2, dtype=torch.float16, device="cuda")

    def check(seq):
        m = min(len(seq), len(seq_ar))
        diff = [i for i in range(m) if seq[i] != seq_ar[i]]
        assert not diff or (diff[0] > len(prompt) + 4 and _near_tie(lm, seq[:diff[0]], seq[diff[0]], seq_ar[diff[0]])), diff[:3]

    model = build()
    out = model.generate(ids, generation_config=gcfg)
    check(out.output_ids[0])
    assert model.lookup_stats["tree"][0] > 0, "the EAGLE tree path never ran"
    model2 = build()
    model2.gen_config = gcfg
    got = list(prompt)
    for new_ids, _ in model2._run_granular(ids, gcfg, 16):
        got.extend(new_ids)
    check(got)
    model3 = build()
    model3.set_cache(gcfg)
    rep = model3.engine.start(ids)
    d = model3.draft.session().read_draft()
    if rep.type == 1 and d.n == 26:
        assert list(d.parent[:26]) == StaticDraftTree(cfg.tree).parents


def test_draft_head_on_device_matches_the_pytorch_head():
    """device_head.DeviceHead (the EAGLE head on the library's kernels, stateless tree levels) against Eagle2Head.forward:
    accepted-token extension in two calls, then two tree levels; output states and logits agree to fp16 tolerance, and
    both plugins (EAGLE-2 dynamic tree, EAGLE v1 static tree) stay lossless when they draft through it."""
    import samd as S
    import samd_sam_only as SO
    from samd.tree_model.device_head import DeviceHead
    from samd.tree_model.eagle import Eagle, EagleHead, StaticDraftTree
    from samd.tree_model.eagle2 import Eagle2, Eagle2Head
    lm = tiny_llama(2, seed=8)
    runner = LlamaRunner.from_hf(lm, max_cache_len=512, dtype=torch.float16)
    tree_cfg = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, vocab_size=512,
                    rms_norm_eps=1e-5, bias=True)
    head = Eagle2Head(tree_cfg, dtype=torch.float16, device="cuda")
    head.random_init(seed=3, std=0.08)
    dh = DeviceHead(head, runner)
    g = torch.Generator(device="cuda").manual_seed(0)
    lm_head = runner.w["lm_head"]
    close = lambda a, b, tol: (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item())
    past = None
    for T in (70, 3):                                            # > 64: the extension runs in two chunks
        hs = torch.randn((T, 256), generator=g, device="cuda").half()
        ids = torch.randint(3, 512, (T,), generator=g, device="cuda")
        out, past = head.forward(hs, ids, past=past)
        last, logits = dh.extend(hs, ids)
        assert close(last, out[-1:], 2e-2) and close(logits, torch.nn.functional.linear(out[-1:], lm_head), 2e-2)
    # two tree levels of 8 nodes: level 1 nodes hang off nodes 2 and 5 of level 0
    L = past[0].shape[1]
    ids0 = torch.randint(3, 512, (8,), generator=g, device="cuda")
    h0 = torch.randn((8, 256), generator=g, device="cuda").half()
    eye = torch.eye(8, device="cuda")
    out0, past1 = head.forward(h0, ids0, past=past, position_ids=torch.full((8,), L, device="cuda"), tree_mask=eye)
    x0 = dh._x(ids0, h0)
    d_out0, d_log0 = dh.tree(x0, torch.zeros(8, dtype=torch.int32, device="cuda"), eye)
    assert close(d_out0, out0, 2e-2)
    par = torch.tensor([2, 2, 2, 5, 5, 5, 5, 2], device="cuda")
    ids1 = torch.randint(3, 512, (8,), generator=g, device="cuda")
    m1 = torch.cat((eye[par], eye), dim=1)
    out1, _ = head.forward(out0[par], ids1, past=past1, position_ids=torch.full((8,), L + 1, device="cuda"), tree_mask=m1)
    anc = torch.zeros((16, 16), device="cuda"); anc[:8, :8] = eye; anc[8:] = m1
    x1 = torch.cat((x0, dh._x(ids1, d_out0[par].clone())), dim=0)
    d_out1, d_log1 = dh.tree(x1, torch.tensor([0] * 8 + [1] * 8, dtype=torch.int32, device="cuda"), anc)
    err_rows = (d_out1[8:].float() - out1.float()).abs().amax(dim=1) / max(1.0, out1.float().abs().max().item())
    assert close(d_out1[8:], out1, 3e-2), f"level-1 states: per-row error / tolerance scale = {err_rows.tolist()} (usual: 1e-3 .. 4e-3, scripts/probes/head_margin.py)"
    assert close(d_log1[8:], torch.nn.functional.linear(out1, lm_head), 3e-2)
    assert dh.length == 73 and int(dh.L.item()) == 73             # tree levels leave the head's cache length alone

    # end to end through both plugins, base model and head on the same kernels
    rng = np.random.default_rng(9)
    prompt = rng.integers(3, 512, 70).tolist()
    prompt[40:52] = prompt[10:22]
    ids = torch.tensor([prompt], device="cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=40, max_cache_len=512)
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, runner, SO.DraftModel(ar_cfg, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
    seq_ar = ar.generate(ids, generation_config=gcfg).output_ids[0]
    for method in ("eagle2", "eagle"):
        cfg = S.SamdConfig(n_predicts=12, len_threshold=3, len_bias=0, tree_method=method, tree_config=tree_cfg)
        if method == "eagle2":
            tm = Eagle2(cfg, runner, torch.float16, "cuda", head=head)
        else:
            h1 = EagleHead(tree_cfg, dtype=torch.float16, device="cuda")
            h1.random_init(seed=3, std=0.08)
            h1.set_tree(StaticDraftTree(cfg.tree))
            tm = Eagle(cfg, runner, torch.float16, "cuda", head=h1)
        assert tm.device_head is not None
        model = S.SamdModel(cfg, runner, S.DraftModel(cfg, tree_model=tm, lm=runner, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
        seq = model.generate(ids, generation_config=gcfg).output_ids[0]
        assert model.lookup_stats["tree"][0] > 0
        if method == "eagle":
            # the static-tree plugin must install ITS tree (EAGLE-2's device-to-device step path builds a dynamic one)
            assert tm.gen_draft_from_step(None, None, 1, 1) is None
            d = model.engine.session.read_draft()
            if d.type == 1 and d.n == len(tm.tree.parents):
                assert list(d.parent[:d.n])[1:] == [int(p) for p in tm.tree.parents[1:]]
        m = min(len(seq), len(seq_ar))
        diff = [i for i in range(m) if seq[i] != seq_ar[i]]
        assert not diff or (diff[0] > len(prompt) + 4 and _near_tie(lm, seq[:diff[0]], seq[diff[0]], seq_ar[diff[0]])), (method, diff[:3])


def test_bf16_gqa_runner_matches_hf():
    """Llama-3-style numerics path: bf16, grouped-query attention (2 query heads per KV head), llama3 rope scaling.
    Tolerance: bf16 has 8 mantissa bits; logits of this 2-layer model agree with fp32 HF to 0.15 absolute."""
    from transformers import LlamaConfig, LlamaForCausalLM, DynamicCache
    torch.manual_seed(11)
    rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=64,
              rope_theta=500000.0)
    try:
        cfg = LlamaConfig(hidden_size=512, intermediate_size=768, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          vocab_size=640, max_position_embeddings=512, head_dim=128, rms_norm_eps=1e-5, tie_word_embeddings=False,
                          rope_parameters=rs)
    except Exception as e:
        pytest.skip(f"LlamaConfig(rope_parameters=...) unsupported: {e}")
    cfg._attn_implementation = "eager"
    lm = LlamaForCausalLM(cfg).cuda().float().eval()
    runner = LlamaRunner.from_hf(lm, max_cache_len=256, dtype=torch.bfloat16)
    sess = samd_hip.Session(512)
    rng = np.random.default_rng(3)
    prompt = rng.integers(3, 640, 90).tolist()
    ids = torch.tensor([prompt], device="cuda")
    last = runner.prefill(sess, ids)
    with torch.no_grad():
        ref = lm(input_ids=ids).logits[0, -1]
    err = (last.float() - ref).abs().max().item()
    assert err < 0.15, err
    assert int(last.float().argmax()) == int(ref.argmax()) or (ref.topk(2).values[0] - ref.topk(2).values[1]).item() < 0.1


def test_token_recycle_on_real_logits_is_lossless():
    """samd[Token Recycle] (BASELINE config 3 loop) on a tiny Llama: the [V,8] table is fed from real fp16 logits inside
    the step's hipGraph; the output must still be the autoregressive greedy sequence."""
    import samd as S
    import samd_sam_only as SO
    lm = tiny_llama(2, seed=9)
    rng = np.random.default_rng(8)
    prompt = rng.integers(3, 512, 48).tolist()
    ids = torch.tensor([prompt], device="cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=64, max_cache_len=512)
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, lm, SO.DraftModel(ar_cfg, device="cuda"), eos_token_id=2, dtype=torch.float16, device="cuda")
    seq_ar = ar.generate(ids, generation_config=gcfg).output_ids[0]
    cfg = S.SamdConfig(n_predicts=16, len_threshold=4, len_bias=0, tree_method="token_recycle")
    draft = S.DraftModel(cfg, lm=lm, device="cuda")
    model = S.SamdModel(cfg, lm, draft, eos_token_id=2, dtype=torch.float16, device="cuda")
    for rep in range(2):                              # the table persists: the second request drafts from what it learned
        out = model.generate(ids, generation_config=gcfg)
        seq = out.output_ids[0]
        m = min(len(seq), len(seq_ar))
        diff = [i for i in range(m) if seq[i] != seq_ar[i]]
        assert not diff or (diff[0] > len(prompt) + 4 and _near_tie(lm, seq[:diff[0]], seq[diff[0]], seq_ar[diff[0]])), diff[:3]
    assert out.decode_steps < out.decode_tokens, "Token Recycle never got a draft accepted on a repeated request"
    assert len(draft.tree_model.cache) > 0


def test_wide_prefill_matches_chunked_and_hf(monkeypatch):
    """prompts of >= 128 tokens go through the one-pass prefill (library GEMMs + samd_prefill_attention / samd_prefill_attention_vt, or fused
    causal SDPA with SAMD_PREFILL_ATTENTION=sdpa); K/V cache, last logits and the following decode must agree with the chunked path and with HF
    (fp16 tolerance 3e-2); the transposed V cache (round 6: the default of "split" as well as "block") must hold exactly what the row-major one
    (SAMD_V_LAYOUT=rows) holds when both run the same attention."""
    lm = tiny_llama(2, seed=13)
    rng = np.random.default_rng(13)
    prompt = rng.integers(3, 512, 200).tolist()
    ids = torch.tensor([prompt], device="cuda")
    outs = {}
    for mode in ("wide", "chunked", "wide-block", "wide-sdpa", "wide-rows", "wide-rows-sdpa"):
        monkeypatch.setenv("SAMD_PREFILL", mode.split("-")[0])
        monkeypatch.setenv("SAMD_PREFILL_ATTENTION", "sdpa" if mode.endswith("sdpa") else "own")
        monkeypatch.setenv("SAMD_V_LAYOUT", "rows" if "rows" in mode else "t")
        runner = LlamaRunner.from_hf(lm, max_cache_len=512, dtype=torch.float16, attention="block" if mode.endswith("block") else "split")
        assert runner.v_transposed == ("rows" not in mode)
        sess = samd_hip.Session(1024)
        last = runner.prefill(sess, ids)
        torch.cuda.synchronize()
        outs[mode] = (last.float().clone(), torch.stack([t.float() for t in runner.kv_rows(200)]), sess.get_cache_length())
        # the same prompt with a per-chunk consumer (what Token Recycle / EAGLE see): all logits and last hidden states
        seen = []
        sess2 = samd_hip.Session(1024)
        runner.prefill(sess2, ids, lambda t, lg, n, h: seen.append((t[:n].clone(), lg[:n].float().clone(), h[:n].float().clone())))
        torch.cuda.synchronize()
        outs[mode] += (torch.cat([x[0] for x in seen]), torch.cat([x[1] for x in seen]), torch.cat([x[2] for x in seen]))
    with torch.no_grad():
        ref = lm(input_ids=ids).logits[0, -1]
    assert outs["wide"][2] == outs["chunked"][2] == 200
    assert (outs["wide"][0] - ref).abs().max().item() < TOL and (outs["chunked"][0] - ref).abs().max().item() < TOL
    assert (outs["wide"][1] - outs["chunked"][1]).abs().max().item() < 2e-2          # same K/V rows up to fp16 GEMM rounding
    assert torch.equal(outs["wide-rows"][1], outs["wide"][1]) and torch.equal(outs["wide-rows"][0], outs["wide"][0])     # the transposed V cache holds the same values
    assert torch.equal(outs["wide-rows-sdpa"][1], outs["wide-sdpa"][1]) and torch.equal(outs["wide-block"][1], outs["wide"][1])
    assert (outs["wide"][1] - outs["wide-sdpa"][1]).abs().max().item() < 2e-2         # our attention kernel / fused SDPA: fp16 roundings
    with torch.no_grad():
        full = lm(input_ids=ids, output_hidden_states=True)
    for mode in outs:
        toks, logits, hidden = outs[mode][3:]
        assert toks.tolist() == prompt and logits.shape == (200, 512) and hidden.shape == (200, 256)
        assert (logits - full.logits[0]).abs().max().item() < TOL
    assert (outs["wide"][5] - outs["chunked"][5]).abs().max().item() < 3e-2          # final-norm hidden states of every prompt token


@pytest.mark.parametrize("impl,bool_mask", [("eager", False), ("sdpa", False), ("sdpa", True)])
def test_attn_patch_dict_forward_is_a_drop_in_for_hf_attention(impl, bool_mask):
    """samd_sam_only.model_patch.attn_patch_dict (the reference's patch registry, model_patch/__init__.py:1-7): binding its
    LlamaAttention.forward keeps an HF model + SamdStaticCache producing HF's logits while a <= 64-row tree step runs on
    samd_tree_attention (fp16 tolerance 3e-2 against the unpatched fp32 model).  With the HF default implementation (sdpa)
    transformers hands the prompt pass NO mask (is_causal is left to SDPA): the patched forward must dispatch that pass the way HF
    does, not to a mask-less eager product; the tree step's mask may also be boolean (True = attend)."""
    import types
    from transformers import DynamicCache
    from transformers.models.llama.modeling_llama import LlamaAttention
    from samd_sam_only.cache import SamdStaticCache
    from samd_sam_only.model_patch import attn_patch_dict
    lm32 = tiny_llama(2, seed=21)
    lm = tiny_llama(2, seed=21).half()
    lm.config._attn_implementation = impl
    rng = np.random.default_rng(21)
    prompt = rng.integers(3, 512, 50).tolist()
    n, L = 19, len(prompt)
    anc = random_parents(rng, n, "bushy")
    toks = rng.integers(3, 512, n).tolist()
    depth = [0] * n
    for i in range(1, n):
        depth[i] = depth[anc[i]] + 1
    ids = torch.tensor([prompt], device="cuda")
    def mask4d(dtype):
        m = torch.full((1, 1, n, L + n), torch.finfo(dtype).min, device="cuda", dtype=dtype)
        m[..., :L] = 0
        for i in range(n):
            j = i
            while j != -1:
                m[0, 0, i, L + j] = 0
                j = anc[j]
        return m
    with torch.no_grad():
        ref_cache = DynamicCache()
        lm32(input_ids=ids, past_key_values=ref_cache, use_cache=True)
        want = lm32(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[L + d for d in depth]], device="cuda"),
                    attention_mask=mask4d(torch.float32), past_key_values=ref_cache, use_cache=True).logits[0]
    name, fn = attn_patch_dict[LlamaAttention][0]
    calls = []
    for mod in lm.modules():
        if isinstance(mod, LlamaAttention):
            def bound(self, *a, _fn=fn, **kw):
                calls.append(a[0].shape[1] if a else kw["hidden_states"].shape[1])
                return _fn(self, *a, **kw)
            setattr(mod, name, types.MethodType(bound, mod))
    cache = SamdStaticCache(lm.config, batch_size=1, max_cache_len=256, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        lm(input_ids=ids, past_key_values=cache, use_cache=True, cache_position=torch.arange(L, device="cuda"))
        cache.set_length()
        got = lm(input_ids=torch.tensor([toks], device="cuda"), position_ids=torch.tensor([[L + d for d in depth]], device="cuda"),
                 attention_mask=(mask4d(torch.float16) == 0) if bool_mask else mask4d(torch.float16), past_key_values=cache, use_cache=True,
                 cache_position=torch.arange(L, L + n, device="cuda")).logits[0]
    assert calls and (got.float() - want).abs().max().item() < TOL


def test_release_row_major_keeps_the_forward(monkeypatch):
    """LlamaRunner.release_row_major (SAMD_RELEASE_ROW_MAJOR=1): the row-major projection matrices go away (memory_report), the prompt
    runs through the streaming kernels in 64-row chunks, and prefill + verify give what the same runner gave with SAMD_PREFILL=chunked
    before the release -- bit for bit, the launches are the same."""
    from samd_hip.llama import LlamaRunner
    cfg = dict(hidden_size=1024, intermediate_size=2816, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=8, vocab_size=4096,
               max_position_embeddings=1024, rms_norm_eps=1e-6)
    monkeypatch.setenv("SAMD_PREFILL", "chunked")
    monkeypatch.setenv("SAMD_QKV_FUSED", "force")                    # (a 1024-wide model has too few q|k|v tiles for the fused forms by default: the
    runner = LlamaRunner.random_init(cfg, 1024, torch.float16, seed=2)   #  norm-fold forward and its group-major o / down copies are what a 7B runner has)
    assert runner.norm_fold
    before = runner.memory_report()
    prompt = torch.tensor([np.random.default_rng(8).integers(3, 4096, 300).tolist()], device="cuda")

    def run():
        sess = samd_hip.Session(1024)
        last = runner.prefill(sess, prompt).float().clone()
        n = 11
        sess.set_draft(torch.arange(5, 5 + n, dtype=torch.int32, device="cuda"), torch.tensor([-1] + list(range(n - 1)), dtype=torch.int32, device="cuda"), n, type_=1)
        b = runner.verify(sess, runner.bucket(n))
        torch.cuda.synchronize()
        return last, b["logits"][:n].float().clone()
    a_last, a_tree = run()
    assert runner.release_row_major() and runner.row_major_released
    after = runner.memory_report()
    keep = 2 * 4096 * 1024 * 2 + 2 * 2 * 1024 * 2                    # the embedding table, lm_head and the layers' norm weights stay
    # round 6: the memory-first mode also keeps ONE packed copy of o_proj / down_proj (the group-major one streams at every row bucket)
    dropped = before["packed_wo"] + before["packed_wdown"]
    assert dropped > 0 and after["packed_wo"] == 0 and after["packed_wdown"] == 0 and after["packed_wo_g"] == before["packed_wo_g"] > 0
    assert after["row_major"] == keep and before["row_major"] > 4 * keep and after["total"] == before["total"] - (before["row_major"] - keep) - dropped
    monkeypatch.delenv("SAMD_PREFILL")
    b_last, b_tree = run()
    assert torch.equal(a_last, b_last) and torch.equal(a_tree, b_tree)
    # ... and it says so where drafts are sized: 64 nodes, not 128 (the 128-row bucket is a library GEMM on the row-major matrices)
    assert runner.max_draft_rows() == 64
    wide = 40                                                         # a 33..48-node draft: the split-K kernel over the group-major copy
    sess = samd_hip.Session(1024)
    runner.prefill(sess, prompt)
    sess.set_draft(torch.arange(5, 5 + wide, dtype=torch.int32, device="cuda"), torch.tensor([-1] + list(range(wide - 1)), dtype=torch.int32, device="cuda"), wide, type_=1)
    got = runner.verify(sess, runner.bucket(wide))["logits"][:wide].float().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and (got[:11] - b_tree).abs().max().item() < 0.05 * max(1.0, b_tree.abs().max().item())
