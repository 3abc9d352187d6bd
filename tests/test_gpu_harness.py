"""The reference's callers on the GPU: the Spec-Bench harness (`run_eval` with the reference's
`sam_only_forward` wrapper, evaluation/inference_sam_only.py:12-33, eval_vicuna.py:71-244), `stream_generate` / the chat session
(samd_model.py:239-285, inference/cli.py) and the sampling branch of eval_posterior (utils.py:142-184) -- with the real decode
engine and LM kernels underneath (a small random-init Llama on the library's runner) and a toy whitespace tokenizer."""
import json
import os
import random

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_harness_cpu import ToyTokenizer        # noqa: E402


MCFG = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
            vocab_size=512, max_position_embeddings=512, rms_norm_eps=1e-5)


def models(seed=4):
    import samd_sam_only as SO
    from samd_hip.llama import LlamaRunner
    runner = LlamaRunner.random_init(MCFG, 512, torch.float16, seed=seed, std=0.05)
    rng = np.random.default_rng(3)
    docs = [rng.integers(20, 200, 40).tolist() for _ in range(30)]
    cfg = SO.SamdConfig(max_predicts=16, alpha=4.0, len_bias=0)
    spec = SO.SamdModel(cfg, runner, SO.DraftModel(cfg, sam_static=SO.build_sam(docs, 2), device="cuda"), 2, torch.float16, "cuda")
    ar_cfg = SO.SamdConfig(max_predicts=1)
    ar = SO.SamdModel(ar_cfg, runner, SO.DraftModel(ar_cfg, device="cuda"), 2, torch.float16, "cuda")
    return spec, ar


class GpuToyTokenizer(ToyTokenizer):
    """ids stay below the runner's vocabulary; unknown generated ids decode to a placeholder word"""

    def __call__(self, texts, return_tensors=None):
        e = super().__call__(texts, return_tensors)
        ids = e.input_ids
        e.input_ids = (ids.cuda() if torch.is_tensor(ids) else ids)
        e.to = lambda device: e
        return e

    def decode(self, ids, **kw):
        return " ".join(self.inv.get(int(i), f"t{int(i)}") for i in ids if int(i) not in (0, 2))


def test_run_eval_on_the_gpu_records_what_generate_returns(tmp_path):
    """run_eval's answer records carry generate()'s own (new tokens, steps, accept lengths) for every turn; the speculative
    run and the autoregressive run of the same kernels give the same answers (evaluation/equal.py) and speed() prices them."""
    from evaluation import equal, run_eval, speed
    from evaluation.inference_sam_only import sam_only_forward
    import samd_sam_only as SO
    spec, ar = models()
    tok = GpuToyTokenizer()
    qs = [{"question_id": 2, "category": "qa", "turns": ["alpha beta gamma delta alpha beta gamma", "and then alpha beta"]},
          {"question_id": 1, "category": "summarization", "turns": ["one two three one two three one two"]}]
    qfile = tmp_path / "q.jsonl"
    qfile.write_text("".join(json.dumps(q) + "\n" for q in qs))
    fwd = lambda inputs, model, tokenizer, max_new_tokens, **kw: sam_only_forward(inputs, model, tokenizer, max_new_tokens, max_cache_len=512)
    a_spec, a_ar = tmp_path / "spec.jsonl", tmp_path / "ar.jsonl"
    run_eval(spec, tok, fwd, "toy-samd", str(qfile), None, None, str(a_spec), max_new_tokens=24)
    run_eval(ar, tok, fwd, "toy-ar", str(qfile), None, None, str(a_ar), max_new_tokens=24)
    rows = [json.loads(l) for l in a_spec.read_text().splitlines()]
    assert [r["question_id"] for r in rows] == [1, 2]
    # first turn of question 1 again, directly: the record is what generate() returns
    from evaluation import get_conversation_template
    conv = get_conversation_template("vicuna")
    conv.append_message(conv.roles[0], qs[1]["turns"][0]); conv.append_message(conv.roles[1], None)
    ids = tok([conv.get_prompt()], return_tensors="pt").input_ids
    out = spec.generate(ids, generation_config=SO.SamdGenerationConfig(max_new_tokens=24, max_cache_len=512))
    c = rows[0]["choices"][0]
    assert c["new_tokens"][0] == out.decode_tokens and c["decoding_steps"][0] == out.decode_steps
    assert c["accept_lengths"][:out.decode_steps] == out.accepet_length_per_step
    assert sum(out.accepet_length_per_step) == out.decode_tokens and all(t > 0 for t in c["wall_time"])
    assert equal(str(a_spec), str(a_ar), report=False)
    tps, tps0, ratio, acc = speed(str(a_spec), str(a_ar), lambda text: len(text.split()) + 1, report=False)
    assert tps > 0 and tps0 > 0 and ratio > 0 and float(np.mean(acc)) >= 1.0


def test_stream_generate_chunks_end_at_generates_text():
    """samd_model.py:239-285: every yielded text extends the previous one and the last equals generate()'s continuation"""
    import samd_sam_only as SO
    spec, _ = models()
    tok = GpuToyTokenizer()
    ids = tok(["alpha beta gamma delta alpha beta gamma delta alpha"], return_tensors="pt").input_ids
    g = SO.SamdGenerationConfig(max_new_tokens=20, max_cache_len=512)
    want = spec.generate(ids, generation_config=g).output_ids[0][ids.shape[1]:]
    texts = [c["text"] for c in spec.stream_generate(ids, tok, generation_config=g)]
    assert len(texts) >= 1 and all(b.startswith(a) for a, b in zip(texts, texts[1:]))
    assert texts[-1].split()[:len(tok.decode(want).split())] == tok.decode(want).split()


def test_chat_session_streams_through_the_engine():
    """evaluation/chat.py (the reference's inference/cli.py loop): a turn's answer is the streamed continuation, the next turn's
    prompt carries the history"""
    import samd_sam_only as SO
    from evaluation.chat import ChatSession, stream_answer
    spec, _ = models()
    tok = GpuToyTokenizer()
    sess = ChatSession("vicuna")
    g = SO.SamdGenerationConfig(max_new_tokens=12, max_cache_len=512)
    seen = []
    p1 = sess.open_turn("alpha beta gamma")
    a1 = stream_answer(spec, tok, p1, g, seen.append)
    sess.close_turn(a1)
    p2 = sess.open_turn("delta")
    assert a1.strip() in p2 and p2.startswith(p1.split("USER:")[0]) and seen and "".join(seen).strip() == a1.strip()
    ids1 = tok([p1], return_tensors="pt").input_ids
    want = spec.generate(ids1, generation_config=g).output_ids[0][ids1.shape[1]:]
    assert a1.split() == tok.decode(want).split()
    a2 = stream_answer(spec, tok, p2, g, lambda s: None)
    assert isinstance(a2, str)


def test_sampling_posterior_on_device_tensors_matches_reference(golden):
    """the sampling branch in the library's kernel (samd_posterior_sampled) against the recorded reference: same host `random`
    stream (the RNG contract: the k-th uniform a step examines is the k-th value random.random() would have returned, one per
    distinct token tried, utils.py:165) -> same accepted prefix and candidate, residual distribution within 1e-5 of the recorded
    fp32 one (device softmax / sum order), and the host generator ends exactly where the reference's loop leaves it."""
    from samd_sam_only import posterior
    from samd_sam_only.utils import SamdGenerationConfig, eval_posterior
    for case in golden("posterior_sampling.json.gz"):
        cfg = SamdGenerationConfig(greedy=False, temperature=case["temperature"], top_p=case["top_p"], top_k=case["top_k"])
        logits = torch.tensor(case["logits"], dtype=torch.float32, device="cuda")
        cand = torch.tensor(case["candidates"], device="cuda")
        random.seed(case["seed"])
        best, acc, sp = eval_posterior(logits, cand, cfg)
        after_device = random.random()
        assert (int(best), int(acc)) == (case["best"], case["accept"])
        assert sp.is_cuda and np.allclose(sp.view(-1).cpu().numpy(), np.asarray(case["sample_p"], dtype=np.float32), atol=1e-5)
        random.seed(case["seed"])
        posterior._sampled(logits.cpu(), cand.cpu(), cfg)                     # the plain restatement of the reference's loop
        assert random.random() == after_device


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sampling_posterior_kernel_follows_the_loop_in_half_precision(dtype):
    """random candidate tries over half-precision logits (what a real model hands over): the kernel and the reference's loop
    (restated in samd_sam_only.posterior._sampled, run on the same device tensors) take the same decisions, leave the host
    generator at the same place and return the same distribution up to one rounding of the dtype."""
    from samd_sam_only import posterior
    from samd_sam_only.utils import SamdGenerationConfig
    g = torch.Generator(device="cuda").manual_seed(3)
    rng = np.random.default_rng(5)
    V = 2000
    for trial in range(12):
        C, D = int(rng.integers(1, 9)), int(rng.integers(2, 7))
        cand = torch.full((C, D), -1, dtype=torch.long)
        cand[:, 0] = 7
        for j in range(C):
            depth = int(rng.integers(2, D + 1))
            cand[j, 1:depth] = torch.from_numpy(rng.integers(3, 12, depth - 1))
        cand = cand.cuda()
        # logits peaked on small token ids so that candidates are accepted and rejected in turn
        logits = torch.randn((C, D, V), generator=g, device="cuda") * 2
        logits[..., 3:12] += 6
        # rows with the same prefix must carry the same logits (they are the same tree node)
        for j in range(C):
            for i in range(D):
                for j0 in range(j):
                    if torch.equal(cand[j0, :i + 1], cand[j, :i + 1]):
                        logits[j, i] = logits[j0, i]
                        break
        logits = logits.to(dtype)
        cfg = SamdGenerationConfig(greedy=False, temperature=float(rng.choice([0.7, 1.0])), top_p=float(rng.choice([0.0, 0.9])), top_k=int(rng.choice([0, 40])))
        random.seed(100 + trial)
        b0, a0, p0 = posterior._sampled(logits, cand, cfg)
        r0 = random.random()
        random.seed(100 + trial)
        b1, a1, p1 = posterior._sampled_device(logits, cand, cfg)
        r1 = random.random()
        assert (int(b0), int(a0), r0) == (int(b1), int(a1), r1), trial
        tol = 2.0 ** (-9 if dtype == torch.float16 else -6)
        assert (p0.float() - p1.float()).abs().max().item() <= tol * max(1e-3, p0.float().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_sampling_posterior_per_node_rows_equal_the_gathered_form(dtype):
    """eval_posterior_nodes (one warped row per draft node, cells read through the retrieve table -- samd_posterior_sampled_nodes) takes
    the decisions of eval_posterior on the gathered [leaves, depth, V] logits the reference builds (samd_model.py:140-146), incl.
    -1 retrieve entries (last node's row, pad token 0) and sequence drafts; same generator position, same distribution."""
    from samd_sam_only.sam.static_sam import gen_buffers
    from samd_sam_only.utils import SamdGenerationConfig, eval_posterior, eval_posterior_nodes
    from util import random_parents
    rng = np.random.default_rng(17)
    g = torch.Generator(device="cuda").manual_seed(17)
    V = 3000
    for trial in range(10):
        n = int(rng.integers(2, 40))
        seq = trial % 4 == 3
        tokens = torch.from_numpy(rng.integers(3, 14, n)).cuda()
        node_logits = torch.randn((n, V), generator=g, device="cuda") * 2
        node_logits[:, 3:14] += 5
        node_logits = node_logits.to(dtype)
        if seq:
            retrieve, cand = None, tokens.view(1, n)
            gathered = node_logits.unsqueeze(0)
        else:
            retrieve = gen_buffers(random_parents(rng, n, ["bushy", "random", "star"][trial % 3]))["tree_retrieve_indices"]
            cand = torch.cat((tokens, torch.zeros(1, dtype=torch.long, device="cuda")))[retrieve]
            gathered = node_logits[retrieve]
        cfg = SamdGenerationConfig(greedy=False, temperature=0.8, top_p=float(rng.choice([0.0, 0.9])), top_k=int(rng.choice([0, 50])))
        random.seed(500 + trial)
        b0, a0, p0 = eval_posterior(gathered, cand, cfg)
        r0 = random.random()
        random.seed(500 + trial)
        b1, a1, p1 = eval_posterior_nodes(node_logits, retrieve, cand, cfg)
        r1 = random.random()
        assert (int(b0), int(a0), r0) == (int(b1), int(a1), r1), trial
        assert torch.equal(p0, p1), trial
