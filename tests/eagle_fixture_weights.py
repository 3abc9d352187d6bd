"""Seeded weights of the head_dim-128 EAGLE fixtures (tests/golden/eagle2_hd128.npz, eagle_hd128.npz).  The generator
(tests/golden/make_golden_eagle_hd128.py, which runs the imported reference) and the GPU tests both call this, so the
fixtures hold only the seed, the inputs and the reference's outputs -- not a megabyte of random numbers.  Every value is
fp16-representable, so the fp32 reference and the fp16 device head see the same weights."""
import numpy as np

CFG = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
           num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-6, pad_token_id=0)

_SHAPES = [("embed_tokens.weight", (512, 256), 0.04), ("fc.weight", (256, 512), 0.04), ("fc.bias", (256,), 0.04),
           ("layers.0.self_attn.q_proj.weight", (256, 256), 0.04), ("layers.0.self_attn.k_proj.weight", (256, 256), 0.04),
           ("layers.0.self_attn.v_proj.weight", (256, 256), 0.04), ("layers.0.self_attn.o_proj.weight", (256, 256), 0.04),
           ("layers.0.mlp.gate_proj.weight", (512, 256), 0.04), ("layers.0.mlp.up_proj.weight", (512, 256), 0.04),
           ("layers.0.mlp.down_proj.weight", (256, 512), 0.04)]


def head_state(seed):
    """name -> float32 array (values rounded to fp16) in the reference's state_dict naming (eagle2_model.py:612-637)"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, std in _SHAPES:
        out[name] = (rng.standard_normal(shape) * std).astype(np.float16).astype(np.float32)
    out["layers.0.post_attention_layernorm.weight"] = (1.0 + 0.05 * rng.standard_normal(256)).astype(np.float16).astype(np.float32)
    return out


def lm_head_weight(seed, std=0.6):
    rng = np.random.default_rng(seed + 7919)
    return (rng.standard_normal((512, 256)) * std).astype(np.float16).astype(np.float32)


def call_inputs(seed, ci, t):
    """hidden states [t, 256] (fp16-representable, |x| ~ 1) and t + 1 token ids of one recorded call"""
    rng = np.random.default_rng(seed * 1000 + 17 * ci + 3)
    hs = rng.standard_normal((t, 256)).astype(np.float16).astype(np.float32)
    ids = rng.integers(3, 512, t + 1).astype(np.int64)
    return hs, ids
