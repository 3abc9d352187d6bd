"""GPU parity of the drop-in API layer (samd_sam_only / samd packages): the reference's own classes and call
sequence (SamdConfig -> DraftModel -> SamdModel.generate) must reproduce, integer for integer, the traces recorded
from the imported reference (tests/golden/loop_so.json.gz, loop_s.json.gz; generator: tests/golden/make_golden.py).
The LM is the scripted verifier (device twin of tests/scripted_lm.py), so every accepted token, step count and
accept length is exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip.engine import ScriptedVerifier


def so_model(case, with_logits=False, use_graphs=True):
    import samd_sam_only as SO
    cfg = SO.SamdConfig(max_predicts=case["max_predicts"], alpha=case["alpha"], K=case["K"], len_bias=case["len_bias"])
    sam = SO.build_sam(case["docs"], case["eos"])
    draft = SO.DraftModel(cfg, sam_static=sam, device="cuda")
    lm = ScriptedVerifier(case["target"], case["vocab"], with_logits=with_logits)
    model = SO.SamdModel(cfg, lm, draft, case["eos"], torch.float16, "cuda")
    return SO, model


@pytest.mark.parametrize("use_graphs", [True, False])
def test_generate_so_matches_reference_trace(golden, use_graphs):
    for case in golden("loop_so.json.gz"):
        SO, model = so_model(case)
        gcfg = SO.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
        ids = torch.tensor([case["prompt"]], dtype=torch.long, device="cuda")
        for rep in range(2):                     # second call: draft.reset() + cache reuse
            model.set_cache(gcfg)
            model.engine.use_graphs = use_graphs
            out = model.generate(ids, generation_config=gcfg)
            assert out.output_ids == [case["output_ids"]]
            assert out.decode_tokens == case["decode_tokens"]
            assert out.decode_steps == case["decode_steps"]
            assert out.accepet_length_per_step == case["accept_lengths"]
        kinds = [t["type"] for t in case["trace"]]
        assert model.lookup_stats["sequence"][0] == 2 * kinds.count("sequence")
        assert model.lookup_stats["tree"][0] == 2 * kinds.count("tree")


def test_max_predicts_above_128_is_capped_not_refused(golden):
    """the reference takes any max_predicts (samd_sam_only/sam/static_sam.py:183); here a draft holds at most 128 nodes (round 5; 64 before),
    so a larger value is served with 128-node drafts: a warning, the same output tokens as max_predicts = 128 step for step, and the same
    tokens as the recorded reference run (speculative decoding is lossless: only accept lengths could differ, and only on matches longer
    than 31).  Values up to 128 are served exactly: tests/test_gpu_wide_drafts.py follows the reference's recorded traces at 80 / 100 / 128."""
    import samd_sam_only as SO
    import samd as S
    case = golden("loop_so.json.gz")[0]
    outs = []
    for mp in (128, 200):
        c = dict(case, max_predicts=mp)
        if mp > 128:
            with pytest.warns(RuntimeWarning, match="capped at 128"):
                _, model = so_model(c)
        else:
            _, model = so_model(c)
        gcfg = SO.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
        model.set_cache(gcfg)
        out = model.generate(torch.tensor([case["prompt"]], dtype=torch.long, device="cuda"), generation_config=gcfg)
        outs.append((out.output_ids, out.accepet_length_per_step))
        assert max(out.accepet_length_per_step) <= 128
    assert outs[0] == outs[1]
    assert outs[1][0] == [case["output_ids"]]                                     # the reference's tokens
    with pytest.warns(RuntimeWarning, match="capped at 128"):
        cfg = S.SamdConfig(n_predicts=190)
    assert cfg.n_predicts == 190
    with pytest.raises(ValueError):
        SO.SamdConfig(max_predicts=0)


def test_granular_decode_so_matches_reference_trace(golden):
    """prefill()/decode()/update_state() with the reference's intermediate tensors."""
    for case in golden("loop_so.json.gz")[:3]:
        SO, model = so_model(case, with_logits=True)
        gcfg = SO.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
        ids = torch.tensor([case["prompt"]], dtype=torch.long, device="cuda")
        model.gen_config = gcfg
        steps = list(model._run_granular(ids, gcfg, gcfg.max_new_tokens))
        got = list(case["prompt"])
        for new_ids, _ in steps:
            got.extend(new_ids)
        assert got[:len(case["prompt"]) + case["max_new_tokens"]] == case["output_ids"]
        assert [len(s[0]) for s in steps] == case["accept_lengths"]


def test_draft_model_api_so(golden):
    """DraftModel.lookup/update return the reference's tuples (type, tokens, buffers)."""
    import samd_sam_only as SO
    case = golden("loop_so.json.gz")[1]
    cfg = SO.SamdConfig(max_predicts=case["max_predicts"], alpha=case["alpha"], K=case["K"], len_bias=case["len_bias"])
    draft = SO.DraftModel(cfg, sam_static=SO.build_sam(case["docs"], case["eos"]), device="cuda")
    draft.reset()
    draft.update(tokens=torch.tensor(case["prompt"]))
    from oracle import sam_oracle as O
    for st in case["trace"]:
        ty, tokens, buf = draft.lookup(st["tokens"][0])
        assert ty.value == st["type"] and tokens == st["tokens"]
        if st["type"] == "sequence":
            assert buf["seq_position_ids"].tolist() == [list(range(len(tokens)))]
        else:
            want = O.gen_buffers(st["anc"])
            assert buf["tree_attn_mask"].cpu().numpy().tolist() == want["tree_attn_mask"].tolist()
            assert buf["tree_position_ids"].cpu().numpy().tolist() == want["tree_position_ids"].tolist()
            assert buf["tree_retrieve_indices"].cpu().numpy().tolist() == want["tree_retrieve_indices"].tolist()
        draft.update(tokens=torch.tensor(st["accepted"]))


def s_model(case):
    import samd as S
    cfg = S.SamdConfig(n_predicts=case["n_predicts"], len_threshold=case["len_threshold"], len_bias=case["len_bias"],
                       tree_method="token_recycle", tree=case["tree"])
    st = S.build_sam(case["docs"], case["eos"]) if case["use_static"] else None
    lm = ScriptedVerifier(case["target"], case["vocab"], with_logits=True)
    draft = S.DraftModel(cfg, sam_static=st, lm=lm, device="cuda")
    model = S.SamdModel(cfg, lm, draft, case["eos"], torch.float16, "cuda")
    return S, model


@pytest.mark.parametrize("use_graphs", [True, False])
def test_generate_s_matches_reference_trace(golden, use_graphs):
    for case in golden("loop_s.json.gz"):
        S, model = s_model(case)
        gcfg = S.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
        model.set_cache(gcfg)
        model.engine.use_graphs = use_graphs
        out = model.generate(torch.tensor([case["prompt"]], dtype=torch.long, device="cuda"), generation_config=gcfg)
        assert out.output_ids == [case["output_ids"]]
        assert out.decode_steps == case["decode_steps"]
        assert out.accepet_length_per_step == case["accept_lengths"]


def test_granular_decode_s_matches_reference_trace(golden):
    case = golden("loop_s.json.gz")[0]
    S, model = s_model(case)
    gcfg = S.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
    model.gen_config = gcfg
    steps = list(model._run_granular(torch.tensor([case["prompt"]], dtype=torch.long, device="cuda"), gcfg, gcfg.max_new_tokens))
    got = list(case["prompt"])
    for new_ids, _ in steps:
        got.extend(new_ids)
    assert got[:len(case["prompt"]) + case["max_new_tokens"]] == case["output_ids"]
    assert [len(s[0]) for s in steps] == case["accept_lengths"]


def test_sam_facades_roundtrip(tmp_path, golden):
    """StaticSAM / DynSAM stand-alone API + dump_sam/load_sam."""
    import samd_sam_only as SO
    from samd_sam_only.sam import DynSAM, StaticSAM
    from oracle import sam_oracle as O
    rng = np.random.default_rng(5)
    docs = [rng.integers(3, 40, 60).tolist() for _ in range(6)] + [[i] for i in range(40)]
    sam = SO.build_sam(docs, 2)
    ora = O.StaticSAM.build(docs, 2)
    path = str(tmp_path / "s.sam")
    SO.dump_sam(path, sam)
    sam2 = SO.load_sam(path)
    stream = rng.integers(3, 40, 50).tolist()
    for s in (sam, sam2):
        s.reset()
        ora.reset()
        for t in stream:
            assert s.lookup(t) == ora.lookup(t)
            s.transfer_tokens([t]); ora.transfer_tokens([t])
            assert (s.cur_index, s.cur_length) == ora.cursor()
    i, l = ora.cursor()
    if l > 0:
        sam.max_predicts = ora.max_predicts = 20
        tree, buf = sam.gen_draft(i, l, stream[-1])
        otree, oanc = ora.gen_draft_tree(i, l, stream[-1])
        assert tree == otree
        assert buf["tree_retrieve_indices"].cpu().numpy().tolist() == O.gen_buffers(oanc)["tree_retrieve_indices"].tolist()
    dyn, odyn = DynSAM(max_predicts=12, alpha=2.0), O.DynSAM(max_predicts=12, alpha=2.0)
    dyn.reset()
    text = (stream[:20] + stream[5:18]) * 2
    dyn.add_tokens(text); odyn.add_tokens(text)
    assert dyn.input_ids == odyn.export()["text"].tolist()
    assert (dyn.cur_index, dyn.cur_length) == odyn.cursor()
    for t in stream[:10]:
        assert dyn.lookup(t) == odyn.lookup(t)
    i, l = odyn.lookup(stream[7])
    assert dyn.gen_draft(i, l, stream[7])[0] == odyn.gen_draft(i, l, stream[7])
    st = dyn.states
    e = odyn.export()
    assert [s.link for s in st] == e["link"].tolist() and [s.min_endpos for s in st] == e["aux"].tolist()


def test_cache_select_indices():
    from samd_sam_only.cache import SamdStaticCache
    cfg = dict(num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, hidden_size=512, max_position_embeddings=128)
    cache = SamdStaticCache(cfg, batch_size=1, max_cache_len=96, device="cuda", dtype=torch.float16, hf_device_map={})
    g = torch.Generator(device="cuda").manual_seed(0)
    k0 = torch.randn((1, 2, 10, 128), generator=g, device="cuda").half()
    for l in range(3):
        kk, vv = cache.update(k0 + l, k0 - l, l)
        assert kk.shape == (1, 2, 10, 128)
    cache.set_length()
    assert cache.get_seq_length() == 10
    k1 = torch.randn((1, 2, 7, 128), generator=g, device="cuda").half()
    for l in range(3):
        cache.update(k1 + l, k1 - l, l)
    idx = torch.tensor([0, 2, 5], device="cuda")
    cache.select_indices(idx, 3)
    assert cache.cache_length == 13
    for l in range(3):
        assert torch.equal(cache.key_cache[l][0, :, 10:13], (k1 + l)[0][:, [0, 2, 5]])
        assert torch.equal(cache.value_cache[l][0, :, 10:13], (k1 - l)[0][:, [0, 2, 5]])
        assert torch.equal(cache.key_cache[l][0, :, :10], (k0 + l)[0])
    cache.select_indices(None, 2)
    assert cache.cache_length == 15
    cache.reset()
    assert cache.get_seq_length() == 0


@pytest.mark.parametrize("seed", range(12))
def test_generate_random_cases_vs_oracle(seed):
    """whole-loop parity on seeded random corpora / prompts / continuations / hyper-parameters: the fused hipGraph step
    behind SamdModel.generate() against the CPU oracle's restatement of the reference loop (same scripted LM)."""
    import samd_sam_only as SO
    from test_oracle_golden import oracle_generate_so
    from util import markov_stream
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.choice([37 * 2 + 1, 97, 150, 211]))                       # coprime with 37 (scripted LM rows)
    while np.gcd(37, V) != 1:
        V += 1
    docs = [markov_stream(rng, int(rng.integers(30, 120)), vocab=V, succ=int(rng.integers(2, 4)), noise=0.03) for _ in range(int(rng.integers(4, 20)))]
    docs += [[i] for i in range(V)]
    flat = [t for d in docs[:-V] for t in d]
    p = int(rng.integers(0, max(1, len(flat) - 60)))
    prompt = flat[p:p + int(rng.integers(8, 40))] + rng.integers(3, V, int(rng.integers(0, 5))).tolist()
    prompt += prompt[2:2 + int(rng.integers(0, 10))]
    cont = []
    max_new = int(rng.integers(20, 90))
    while len(cont) < max_new + 70:
        r = rng.random()
        if r < 0.5:
            q = int(rng.integers(0, max(1, len(flat) - 30))); cont += flat[q:q + int(rng.integers(3, 25))]
        elif r < 0.75 and len(prompt) > 8:
            q = int(rng.integers(0, len(prompt) - 4)); cont += prompt[q:q + int(rng.integers(2, 9))]
        else:
            cont += rng.integers(3, V, int(rng.integers(1, 4))).tolist()
    cont = [t if t != 2 else 3 for t in cont]
    if seed % 3 == 0:
        cont[int(rng.integers(5, max_new))] = 2                             # an EOS inside the continuation
    case = {"docs": docs, "eos": 2, "vocab": V, "max_predicts": int(rng.choice([1, 4, 16, 40, 60, 64])), "alpha": float(rng.choice([1.0, 2.5, 4.0])),
            "K": int(rng.choice([1, 3, 8])), "len_bias": int(rng.choice([0, 0, 2, 5])), "max_new_tokens": max_new,
            "max_cache_len": int(rng.choice([len(prompt) + 70, 256, 512])), "prompt": prompt, "target": prompt + cont}
    want = oracle_generate_so(case)
    SO_, model = so_model(case)
    gcfg = SO_.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
    out = model.generate(torch.tensor([prompt], dtype=torch.long, device="cuda"), generation_config=gcfg)
    assert out.output_ids == [want["output_ids"]]
    assert (out.decode_tokens, out.decode_steps, out.accepet_length_per_step) == (want["decode_tokens"], want["decode_steps"], want["accept_lengths"])


@pytest.mark.parametrize("seed", range(8))
def test_generate_s_random_cases_vs_oracle(seed, golden):
    """full variant (SAM sequences + Token Recycle trees) on seeded random inputs: samd.SamdModel.generate() (Token-Recycle
    update/fill inside the step's hipGraph) against the oracle loop that is itself pinned by the reference's traces."""
    from test_oracle_golden import oracle_generate_s
    from util import markov_stream
    rng = np.random.default_rng(2000 + seed)
    V = 97 if seed % 2 else 150
    while np.gcd(37, V) != 1:
        V += 1
    docs = [markov_stream(rng, int(rng.integers(40, 100)), vocab=V, succ=2, noise=0.03) for _ in range(int(rng.integers(4, 14)))] + [[i] for i in range(V)]
    flat = [t for d in docs[:-V] for t in d]
    p = int(rng.integers(0, max(1, len(flat) - 50)))
    prompt = flat[p:p + int(rng.integers(10, 36))] + rng.integers(3, V, int(rng.integers(0, 4))).tolist()
    prompt += prompt[3:3 + int(rng.integers(0, 9))]
    max_new = int(rng.integers(24, 70))
    cont = []
    while len(cont) < max_new + 70:
        r = rng.random()
        if r < 0.5:
            q = int(rng.integers(0, max(1, len(flat) - 30))); cont += flat[q:q + int(rng.integers(3, 22))]
        elif r < 0.75 and len(prompt) > 8:
            q = int(rng.integers(0, len(prompt) - 4)); cont += prompt[q:q + int(rng.integers(2, 9))]
        else:
            cont += rng.integers(3, V, int(rng.integers(1, 4))).tolist()
    cont = [t if t != 2 else 3 for t in cont]
    if seed % 4 == 1:
        cont[int(rng.integers(5, max_new))] = 2
    tree = golden("buffers.json.gz")["token_recycle"][0]["tree"]
    case = {"docs": docs, "eos": 2, "vocab": V, "n_predicts": int(rng.choice([6, 12, 40])), "len_threshold": int(rng.choice([2, 3, 5])),
            "len_bias": int(rng.choice([0, 1, 5])), "max_predicts": 70, "use_static": bool(seed % 3), "tree": tree,
            "max_new_tokens": max_new, "max_cache_len": 512, "prompt": prompt, "target": prompt + cont}
    want = oracle_generate_s(case)
    S, model = s_model(case)
    gcfg = S.SamdGenerationConfig(max_new_tokens=max_new, max_cache_len=512)
    out = model.generate(torch.tensor([prompt], dtype=torch.long, device="cuda"), generation_config=gcfg)
    assert out.output_ids == [want["output_ids"]]
    assert (out.decode_steps, out.accepet_length_per_step) == (want["decode_steps"], want["accept_lengths"])
    assert model.lookup_stats["tree"][0] == want["types"].count("tree")


def test_sampling_generate_runs_and_is_seeded():
    """non-greedy generation goes through the granular prefill/decode path with HF's logits warpers and the host RNG
    (samd_sam_only/utils.py:142-184); with the same seeds it is reproducible, and every token is a valid id."""
    import random
    import samd_sam_only as SO
    from samd_hip.llama import LlamaRunner
    mcfg = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                vocab_size=512, max_position_embeddings=256, rms_norm_eps=1e-5)
    prompt = np.random.default_rng(0).integers(3, 512, 20).tolist()
    ids = torch.tensor([prompt], device="cuda")
    g = SO.SamdGenerationConfig(max_new_tokens=24, max_cache_len=256, greedy=False, temperature=0.8, top_p=0.9)
    outs = []
    for rep in range(2):
        runner = LlamaRunner.random_init(mcfg, 256, torch.float16, seed=4, std=0.05)
        cfg = SO.SamdConfig(max_predicts=8, len_bias=0)
        model = SO.SamdModel(cfg, runner, SO.DraftModel(cfg, device="cuda"), 2, torch.float16, "cuda")
        random.seed(7); torch.manual_seed(7)
        outs.append(model.generate(ids, generation_config=g))
    assert outs[0].output_ids == outs[1].output_ids
    seq = outs[0].output_ids[0]
    assert seq[:len(prompt)] == prompt and len(seq) > len(prompt) and all(0 <= t < 512 for t in seq)


@pytest.mark.parametrize("with_static", [False, True])
@pytest.mark.parametrize("temperature,top_p,top_k", [(0.8, 0.9, 0), (1.0, 0.0, 20), (0.3, 0.0, 0)])
def test_fused_sampling_step_equals_the_granular_loop(monkeypatch, with_static, temperature, top_p, top_k):
    """generate(greedy=False) in its fused form (engine.step_sampled: candidates, warpers, the accept / reject walk, the next start
    token's draw and samd_session_step_given all on the device, ONE host synchronisation per step) must reproduce the granular
    prefill() / decode() loop -- the reference's own shape (samd_model.py:96-174, utils.py:66-184), itself pinned to the recorded
    reference at the posterior level -- token for token, step for step, and leave BOTH random streams (host `random`, torch) where that
    loop leaves them.  Sequence drafts only (no static automaton) and tree drafts (a static automaton the prompt was drawn from)."""
    import random
    import samd_sam_only as SO
    from samd_hip.llama import LlamaRunner
    from util import markov_stream
    mcfg = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                vocab_size=512, max_position_embeddings=256, rms_norm_eps=1e-5)
    rng = np.random.default_rng(3)
    docs = [markov_stream(rng, 200, vocab=512) for _ in range(6)] + [[i] for i in range(512)]
    prompt = (docs[1][20:44] + docs[3][5:17])[:32]
    ids = torch.tensor([prompt], device="cuda")
    g = SO.SamdGenerationConfig(max_new_tokens=28, max_cache_len=256, greedy=False, temperature=temperature, top_p=top_p, top_k=top_k)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SAMD_FUSED_SAMPLING", mode)
        runner = LlamaRunner.random_init(mcfg, 256, torch.float16, seed=4, std=0.05)
        cfg = SO.SamdConfig(max_predicts=16, len_bias=0)
        sam = SO.build_sam(docs, 2) if with_static else None
        model = SO.SamdModel(cfg, runner, SO.DraftModel(cfg, sam_static=sam, device="cuda"), 2, torch.float16, "cuda")
        random.seed(11); torch.manual_seed(11)
        out = model.generate(ids, generation_config=g)
        res[mode] = (out.output_ids, out.decode_steps, out.accepet_length_per_step, random.random(), torch.rand(1, device="cuda").item(),
                     dict((k, list(v)) for k, v in model.lookup_stats.items()))
    assert res["1"] == res["0"]
    assert res["1"][1] >= 1 and len(res["1"][0][0]) > len(prompt)
    if with_static:
        assert res["1"][5]["tree"][0] > 0                   # tree drafts took part


def test_runner_shorter_than_generation_config_is_an_error():
    """a ready-made runner holds K/V rows for ITS max_cache_len only: asking for a longer generation must raise instead of
    silently dropping rows past the end (reference: the cache is always sized from generation_config, samd_model.py:176-191);
    a shorter or equal request keeps working on the same runner."""
    import samd_hip
    import samd_sam_only as SO
    from samd_hip.llama import LlamaRunner
    mcfg = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                vocab_size=512, max_position_embeddings=512, rms_norm_eps=1e-5)
    runner = LlamaRunner.random_init(mcfg, 128, torch.float16, seed=4, std=0.05)
    cfg = SO.SamdConfig(max_predicts=8, len_bias=0)
    model = SO.SamdModel(cfg, runner, SO.DraftModel(cfg, device="cuda"), 2, torch.float16, "cuda")
    ids = torch.tensor([np.random.default_rng(1).integers(3, 512, 12).tolist()], device="cuda")
    with pytest.raises(samd_hip.SamdError, match="exceeds the runner's max_cache_len 128"):
        model.generate(ids, generation_config=SO.SamdGenerationConfig(max_new_tokens=8, max_cache_len=256))
    out = model.generate(ids, generation_config=SO.SamdGenerationConfig(max_new_tokens=8, max_cache_len=128))
    assert len(out.output_ids[0]) > ids.shape[1]
    out2 = model.generate(ids, generation_config=SO.SamdGenerationConfig(max_new_tokens=8, max_cache_len=96))
    assert out2.output_ids == out.output_ids
