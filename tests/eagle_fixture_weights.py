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


# ---------------------------------------------------------------------------------------------------------------------
# The PLANTED fixture (tests/golden/eagle2_planted_bf16.npz, round 4): a head whose top-k decisions are DECIDABLE in bf16.
#
# Random weights give a flat next-token distribution, and an EAGLE-2 expansion makes ~300 ordered decisions per call, so some
# decision is always closer than bf16's rounding noise and "identical draft" cannot be demanded (eagle2_hd128_bf16.npz).  A real
# draft head is peaked.  Here the peaks are planted: `slots` are the tokens that get EXPANDED (root + 8 rows x 5 levels per call);
# slot s has an exactly representable embedding ALPHA * q_s (q_s = signed Sylvester-Hadamard rows / 16: orthonormal, entries
# +-1/16), fc passes the embedding through (fc.weight = [I | B]), and lm_head row v of a planted successor of slot s is
# ((L + COLD) / ALPHA) q_s - (COLD / BETA) u, u = the Hadamard row the fc bias carries (BETA * u).  So in the row of slot s the
# successor's logit is ~ L, every other token's is ~ -COLD; the head's attention, MLP and the hidden-state half of fc perturb all
# of that by a few tenths -- they decide near cases, which is what gives the test its power -- and the generator
# (tests/golden/make_golden_eagle_planted.py) tunes the ladders L against the IMPORTED reference until every ordered decision it
# records has a margin >= 10x the bf16 noise.  The fixture stores only the seed and the slot tables; generator and GPU test both
# build the weights here.  All values are representable in bf16.
# ---------------------------------------------------------------------------------------------------------------------
PLANT = dict(hidden=256, inter=512, heads=2, vocab=32000, alpha=4.0, beta=4.0, cold=8.0, ranks=10,
             std_b=0.02, std_layer=0.04, std_cold=0.004, std_embed=0.02)


def _hadamard(n):
    i = np.arange(n, dtype=np.uint32)
    x = i[:, None] & i[None, :]
    pop = np.zeros_like(x)
    while x.any():
        pop += x & 1
        x >>= 1
    return np.where(pop & 1, -1.0, 1.0).astype(np.float32)


def planted_basis(seed):
    """rows 1..255: slot directions q_s; row 0: the bias direction u.  Orthonormal, entries +-1/16."""
    D = PLANT["hidden"]
    rng = np.random.default_rng(seed + 424242)
    d = rng.integers(0, 2, D).astype(np.float32) * 2 - 1
    perm = rng.permutation(np.arange(1, D))
    H = _hadamard(D) * d[None, :] / np.sqrt(D)
    return H[perm], H[0]


def planted_config(vocab=None):
    return dict(vocab_size=vocab or PLANT["vocab"], hidden_size=PLANT["hidden"], intermediate_size=PLANT["inter"], num_hidden_layers=1,
                num_attention_heads=PLANT["heads"], num_key_value_heads=PLANT["heads"], max_position_embeddings=256, rms_norm_eps=1e-6,
                pad_token_id=0)


def planted_head_state(seed, slot_tokens):
    """state dict of the head in the reference's naming; slot_tokens[s] = the token whose embedding is ALPHA * q_s"""
    P, D, I, V = PLANT, PLANT["hidden"], PLANT["inter"], PLANT["vocab"]
    rng = np.random.default_rng(seed)
    q, u = planted_basis(seed)
    out = {}
    emb = rng.standard_normal((V, D)).astype(np.float32) * P["std_embed"]
    st = np.asarray(slot_tokens, dtype=np.int64)
    emb[st] = P["alpha"] * q[:len(st)]
    out["embed_tokens.weight"] = _round(emb, "bf16")
    B = rng.standard_normal((D, D)).astype(np.float32) * P["std_b"]
    out["fc.weight"] = _round(np.concatenate([np.eye(D, dtype=np.float32), B], axis=1), "bf16")
    out["fc.bias"] = _round(P["beta"] * u + rng.standard_normal(D).astype(np.float32) * 0.01, "bf16")
    for name, shape in (("q_proj", (D, D)), ("k_proj", (D, D)), ("v_proj", (D, D)), ("o_proj", (D, D))):
        out[f"layers.0.self_attn.{name}.weight"] = _round(rng.standard_normal(shape).astype(np.float32) * P["std_layer"], "bf16")
    for name, shape in (("gate_proj", (I, D)), ("up_proj", (I, D)), ("down_proj", (D, I))):
        out[f"layers.0.mlp.{name}.weight"] = _round(rng.standard_normal(shape).astype(np.float32) * P["std_layer"], "bf16")
    out["layers.0.post_attention_layernorm.weight"] = _round(1.0 + 0.05 * rng.standard_normal(D).astype(np.float32), "bf16")
    return out


def planted_lm_head(seed, slot_succ, slot_logit):
    """lm_head [V, hidden]: cold rows -(COLD/BETA) u + noise; successor j of slot s gets ((L_sj + COLD)/ALPHA) q_s - (COLD/BETA) u"""
    P, D, V = PLANT, PLANT["hidden"], PLANT["vocab"]
    rng = np.random.default_rng(seed + 7919)
    q, u = planted_basis(seed)
    w = rng.standard_normal((V, D)).astype(np.float32) * P["std_cold"] - (P["cold"] / P["beta"]) * u[None, :]
    succ = np.asarray(slot_succ, dtype=np.int64)
    logit = np.asarray(slot_logit, dtype=np.float32)
    for s in range(succ.shape[0]):
        w[succ[s]] = ((logit[s][:, None] + P["cold"]) / P["alpha"]) * q[s][None, :] - (P["cold"] / P["beta"]) * u[None, :]
    return _round(w, "bf16")


def planted_tokens(seed):
    """the token ids the fixture uses, drawn once: fresh planted tokens (call roots, successors) are taken from the FRONT of this
    permutation in order of allocation, the cold context ids of the calls from its END -- the two never meet"""
    rng = np.random.default_rng(seed + 31337)
    return rng.permutation(np.arange(3, PLANT["vocab"]))


def planted_call_inputs(seed, ci, t, root_token):
    """hidden states [t, hidden] (bf16-representable, |x| ~ 1) and t + 1 token ids: t cold context tokens, then the call's root
    slot token (the token whose successors the expansion starts from)"""
    rng = np.random.default_rng(seed * 1000 + 17 * ci + 5)
    hs = _round(rng.standard_normal((t, PLANT["hidden"])), "bf16")
    perm = planted_tokens(seed)
    ids = perm[len(perm) - 64 * (ci + 1): len(perm) - 64 * (ci + 1) + t].tolist()
    return hs, np.asarray(ids + [int(root_token)], dtype=np.int64)
