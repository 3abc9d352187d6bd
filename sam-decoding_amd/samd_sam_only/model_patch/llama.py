"""Tree-mask attention for the verify forward (reference: samd_sam_only/model_patch/llama.py:35-109, :112-202).

The reference injects the tree mask into HF's 4-D additive causal mask (llama.py:94-96): the n new tokens see all
`cache_length` cached keys and, among the new keys, themselves and their tree ancestors.  Here the same rule is the
u64-row mask consumed by the hand-written kernel (samd_tree_attention); `tree_decode_mask` restates the reference's
additive mask for comparisons against plain PyTorch SDPA.
"""
import math

import torch

import samd_hip


def tree_decode_mask(tree_attn_mask: torch.Tensor, cache_length: int, dtype=torch.float32):
    """[1,1,n,cache_length+n] additive mask: zeros over the cached keys, min_dtype where tree_attn_mask == 0."""
    n = tree_attn_mask.shape[-1]
    m = torch.zeros((1, 1, n, cache_length + n), dtype=dtype, device=tree_attn_mask.device)
    m[..., cache_length:] = torch.finfo(dtype).min * (tree_attn_mask.reshape(1, 1, n, n) == 0)
    return m


def mask_rows_u64(tree_attn_mask: torch.Tensor) -> torch.Tensor:
    """bool/float [.., n, n] mask -> int64[2 * MAX_DRAFT] whose bit patterns are the u64 row masks the kernel reads: the low words (nodes
    0..63) of all rows, then -- MAX_DRAFT entries further -- the high words (nodes 64..127; read only when more than 64 rows are verified)."""
    n = tree_attn_mask.shape[-1]
    bits = (tree_attn_mask.reshape(n, n) != 0).to(torch.int64).cpu()
    M = samd_hip.MAX_DRAFT
    lo = [sum(int(bits[i, j]) << j for j in range(min(n, 64))) for i in range(n)] + [0] * (M - n)
    hi = [sum(int(bits[i, j]) << (j - 64) for j in range(64, n)) for i in range(n)] + [0] * (M - n)
    rows = [r - (1 << 64) if r >= (1 << 63) else r for r in lo + hi]
    return torch.tensor(rows, dtype=torch.int64, device="cuda")


def tree_attention(q, k_cache, v_cache, tree_attn_mask, cache_length: int, n: int, scale=None):
    """q [n, H, D]; k_cache / v_cache [H_kv, max_len, D] with the new rows already written at [cache_length, +n).
    Returns [n, H, D]."""
    H, D = q.shape[1], q.shape[2]
    n_pad = q.shape[0]
    out = torch.empty_like(q)
    ws_bytes = samd_hip.lib().samd_tree_attention_workspace(n_pad, H, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    d_L = torch.tensor([cache_length], dtype=torch.int32, device=q.device)
    d_n = torch.tensor([n], dtype=torch.int32, device=q.device)
    mask = mask_rows_u64(tree_attn_mask)
    samd_hip.check(samd_hip.lib().samd_tree_attention(
        samd_hip._ptr(q), samd_hip._ptr(k_cache), samd_hip._ptr(v_cache), samd_hip._ptr(out), samd_hip.torch_dtype_code(q.dtype),
        n_pad, H, k_cache.shape[0], D, k_cache.shape[1], samd_hip._ptr(mask), samd_hip._ptr(d_L), samd_hip._ptr(d_n),
        scale if scale is not None else 1.0 / math.sqrt(D), samd_hip._ptr(ws), ws_bytes, samd_hip.current_stream()))
    torch.cuda.current_stream().synchronize()      # d_L / d_n / mask / ws are locals
    return out[:n]


def hf_attention_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
    """Drop-in for transformers' LlamaAttention.forward (the reference patches HF modules through `attn_patch_dict`,
    samd_sam_only/model_patch/__init__.py:1-7): same projections, rotary embedding and cache update as HF, but the attention product
    of a draft step -- bs = 1, <= 64 new rows, head_dim 128, fp16 / bf16, K/V views of a row-major SamdStaticCache, a 4-D additive
    mask (additive float, or bool with True = attend) whose new-key block is the tree mask (model_patch/llama.py:94-96) -- runs on
    samd_tree_attention.  Anything else (prefill chunks > 64 rows, no explicit mask, other caches, fp32) is dispatched exactly as HF's
    own forward does: the attention interface the model was configured with (`config._attn_implementation`: sdpa applies its own
    causal rule when transformers dropped the mask; eager only when eager was asked for)."""
    from transformers.models.llama.modeling_llama import ALL_ATTENTION_FUNCTIONS, apply_rotary_pos_emb, eager_attention_forward
    input_shape = hidden_states.shape[:-1]
    hidden_shape = (*input_shape, -1, self.head_dim)
    q = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    k = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    v = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    cos, sin = position_embeddings
    q, k = apply_rotary_pos_emb(q, k, cos, sin)
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    n, total = q.shape[2], k.shape[2]
    D = self.head_dim
    usable = (q.shape[0] == 1 and n <= samd_hip.TILE_ROWS and D == 128 and q.dtype in (torch.float16, torch.bfloat16) and q.is_cuda
              and attention_mask is not None and attention_mask.dim() == 4 and attention_mask.shape[-1] >= total
              and (attention_mask.dtype.is_floating_point or attention_mask.dtype == torch.bool)
              and k.stride(3) == 1 and k.stride(2) == D and v.stride(3) == 1 and v.stride(2) == D and k.stride(1) == v.stride(1)
              and k.stride(1) % D == 0 and not getattr(past_key_values, "v_transposed", False))
    if not usable:
        interface = ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, eager_attention_forward)
        out, w = interface(self, q, k, v, attention_mask, dropout=0.0 if not self.training else self.attention_dropout, scaling=self.scaling, **kwargs)
        return self.o_proj(out.reshape(*input_shape, -1).contiguous()), w
    L = total - n
    H, Hkv, max_len = q.shape[1], k.shape[1], k.stride(1) // D
    blk = attention_mask[0, 0, :n, L:total]
    m = (blk if blk.dtype == torch.bool else blk == 0).to(torch.int64)          # bool: True = attend; additive: 0 = attend
    rows = torch.zeros(samd_hip.MAX_DRAFT, dtype=torch.int64, device=q.device)
    rows[:n] = (m << torch.arange(n, device=q.device, dtype=torch.int64)[None, :]).sum(-1)
    n_pad = n
    qn = q[0].transpose(0, 1).contiguous()                                   # [n, H, D]
    out = torch.empty_like(qn)
    ws_bytes = samd_hip.lib().samd_tree_attention_workspace(n_pad, H, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    d_L = torch.tensor([L], dtype=torch.int32, device=q.device)
    d_n = torch.tensor([n], dtype=torch.int32, device=q.device)
    samd_hip.check(samd_hip.lib().samd_tree_attention(
        samd_hip._ptr(qn), samd_hip._ptr(k), samd_hip._ptr(v), samd_hip._ptr(out), samd_hip.torch_dtype_code(q.dtype), n_pad, H, Hkv, D, max_len,
        samd_hip._ptr(rows), samd_hip._ptr(d_L), samd_hip._ptr(d_n), float(self.scaling), samd_hip._ptr(ws), ws_bytes, samd_hip.current_stream()))
    torch.cuda.current_stream().synchronize()                                 # rows / d_L / d_n / ws are locals
    return self.o_proj(out.reshape(*input_shape, -1)), None


def _runner_for(lm, max_cache_len, dtype, device, **kw):
    from samd_hip.llama import LlamaRunner
    return LlamaRunner.from_hf(lm, max_cache_len, dtype, device, **kw)


def _tables():
    try:
        from transformers import LlamaForCausalLM
        from transformers.models.llama.modeling_llama import LlamaAttention
    except Exception:                                # transformers absent: the runner can still be built from raw weights
        return {}, {}
    # patch_dict: what SamdModel swaps the LM's forward for (its own decoder loop, samd_hip.llama.LlamaRunner);
    # attn_patch_dict: a forward with HF's LlamaAttention signature for callers that keep driving the HF module themselves
    return {LlamaForCausalLM: [("forward", _runner_for)]}, {LlamaAttention: [("forward", hf_attention_forward)]}


patch_dict, attn_patch_dict = _tables()
