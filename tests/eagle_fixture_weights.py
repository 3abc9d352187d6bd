"""Seeded weights of the head_dim-128 EAGLE fixtures (tests/golden/eagle2_hd128.npz, eagle_hd128.npz).  The generator
(tests/golden/make_golden_eagle_hd128.py, which runs the imported reference) and the GPU tests both call this, so the
fixtures hold only the seed, the inputs and the reference's outputs -- not a megabyte of random numbers.  Every value is
representable in the device head's dtype (`rounding` = "f16" or "bf16"), so the fp32 reference and the half-precision device head
see the same weights.  `vocab` widens the embedding table and the lm_head (the V = 32000 fixture); everything else keeps CFG."""
import numpy as np


def _round(x, rounding):
    """float64/32 array -> float32 values representable in fp16 / bf16 (bf16: round-to-nearest-even on the upper 16 bits)"""
    if rounding == "f16":
        return np.asarray(x).astype(np.float16).astype(np.float32)      # straight from float64, as the committed fixtures were made
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)

CFG = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
           num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-6, pad_token_id=0)

_SHAPES = [("embed_tokens.weight", (512, 256), 0.04), ("fc.weight", (256, 512), 0.04), ("fc.bias", (256,), 0.04),
           ("layers.0.self_attn.q_proj.weight", (256, 256), 0.04), ("layers.0.self_attn.k_proj.weight", (256, 256), 0.04),
           ("layers.0.self_attn.v_proj.weight", (256, 256), 0.04), ("layers.0.self_attn.o_proj.weight", (256, 256), 0.04),
           ("layers.0.mlp.gate_proj.weight", (512, 256), 0.04), ("layers.0.mlp.up_proj.weight", (512, 256), 0.04),
           ("layers.0.mlp.down_proj.weight", (256, 512), 0.04)]


def head_state(seed, rounding="f16", vocab=512):
    """name -> float32 array (values rounded to `rounding`) in the reference's state_dict naming (eagle2_model.py:612-637)"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, std in _SHAPES:
        if name == "embed_tokens.weight":
            shape = (vocab, shape[1])
        out[name] = _round(rng.standard_normal(shape) * std, rounding)
    out["layers.0.post_attention_layernorm.weight"] = _round(1.0 + 0.05 * rng.standard_normal(256), rounding)
    return out


def lm_head_weight(seed, std=0.6, rounding="f16", vocab=512):
    rng = np.random.default_rng(seed + 7919)
    return _round(rng.standard_normal((vocab, 256)) * std, rounding)


def call_inputs(seed, ci, t, rounding="f16", vocab=512):
    """hidden states [t, 256] (representable in `rounding`, |x| ~ 1) and t + 1 token ids of one recorded call"""
    rng = np.random.default_rng(seed * 1000 + 17 * ci + 3)
    hs = _round(rng.standard_normal((t, 256)), rounding)
    ids = rng.integers(3, vocab, t + 1).astype(np.int64)
    return hs, ids
