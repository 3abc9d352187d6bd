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
    """bool/float [.., n, n] mask -> int64[64] whose bit pattern is the u64 row mask the kernel reads."""
    n = tree_attn_mask.shape[-1]
    bits = (tree_attn_mask.reshape(n, n) != 0).to(torch.int64).cpu()
    rows = [sum(int(bits[i, j]) << j for j in range(n)) for i in range(n)] + [0] * (samd_hip.MAX_DRAFT - n)
    rows = [r - (1 << 64) if r >= (1 << 63) else r for r in rows]
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


def _runner_for(lm, max_cache_len, dtype, device, **kw):
    from samd_hip.llama import LlamaRunner
    return LlamaRunner.from_hf(lm, max_cache_len, dtype, device, **kw)


def _tables():
    try:
        from transformers import LlamaForCausalLM
        from transformers.models.llama.modeling_llama import LlamaAttention
    except Exception:                                # transformers absent: the runner can still be built from raw weights
        return {}, {}
    return {LlamaForCausalLM: [("forward", _runner_for)]}, {LlamaAttention: [("forward", tree_attention)]}


patch_dict, attn_patch_dict = _tables()
