"""SamdStaticCache / SamdCache -- the KV cache of the verify forward.

Semantics of samd_sam_only/cache.py:37-133: one pre-allocated [1, H_kv, max_cache_len, D] K and V per layer,
`update` writes the new rows at [cache_length, cache_length + n) and returns views of [0, cache_length + n),
`select_indices` keeps the accepted rows (start + idx[j] -> start + j) and advances cache_length, rejected rows are
overwritten by the next step.  Storage is a single HBM allocation [layers, 2, H_kv, max_len, D]; compaction of all
2 x layers tensors is one kernel launch over a pointer table (samd_kv_compact_indices / samd_kv_compact).
When the cache backs a LlamaRunner (SamdModel.set_cache) the V half of every layer is kept TRANSPOSED in the same bytes
([H_kv][D][max_len], the operand layout of samd_attention_block); `value_cache` then holds transposed views, so `update`,
`select_indices` and anything that indexes value_cache[l][0, head, position] behave as before.
The layout is contiguous, not paged: at bs=1 with max_cache_len <= 8192 a request's whole cache is <= 4 GiB of the
288 GB HBM, and contiguous rows keep the attention kernel's K/V tile loads fully coalesced.
"""
from typing import List, Optional

import torch

import samd_hip


def _cfg(config, name, default=None):
    return config.get(name, default) if isinstance(config, dict) else getattr(config, name, default)


class SamdStaticCache:

    def __init__(self, config, batch_size=None, max_cache_len=None, device=None, dtype=torch.float32, max_batch_size=None,
                 hf_device_map=None):
        samd_hip.require_gpu()
        self.batch_size = batch_size or max_batch_size or 1
        assert self.batch_size == 1, "Only support batch_size == 1"
        self.max_cache_len = _cfg(config, "max_position_embeddings") if max_cache_len is None else max_cache_len
        heads = _cfg(config, "num_attention_heads")
        self.head_dim = _cfg(config, "head_dim") or _cfg(config, "hidden_size") // heads
        self.dtype = dtype
        kv = _cfg(config, "num_key_value_heads")
        self.num_key_value_heads = heads if kv is None else kv
        self.num_layers = _cfg(config, "num_hidden_layers")
        dev = torch.device(device if device is not None else "cuda")
        self.storage = torch.zeros((self.num_layers, 2, self.num_key_value_heads, self.max_cache_len, self.head_dim), dtype=dtype, device=dev)
        self.key_cache: List[torch.Tensor] = [self.storage[l, 0].unsqueeze(0) for l in range(self.num_layers)]
        self.v_transposed = False
        self.value_cache: List[torch.Tensor] = [self.storage[l, 1].unsqueeze(0) for l in range(self.num_layers)]
        self._ptrs = torch.tensor([self.storage[l, j].data_ptr() for j in (0, 1) for l in range(self.num_layers)],
                                  dtype=torch.int64, device=dev)
        self.last_length = 0
        self.cache_length = 0

    def set_v_transposed(self, flag=True):
        """keep V as [H_kv][D][max_len] inside the same storage (only on an empty cache)"""
        assert self.cache_length == 0 and self.last_length == 0, "the V layout can only change while the cache is empty"
        self.v_transposed = bool(flag)
        H, L, D = self.num_key_value_heads, self.max_cache_len, self.head_dim
        if flag:
            self.value_cache = [self.storage[l, 1].view(H, D, L).transpose(1, 2).unsqueeze(0) for l in range(self.num_layers)]
        else:
            self.value_cache = [self.storage[l, 1].unsqueeze(0) for l in range(self.num_layers)]

    def reset(self):
        self.cache_length = 0
        self.last_length = 0

    def set_length(self):
        self.cache_length = self.last_length

    def get_seq_length(self, layer_idx=0):
        return self.cache_length

    def get_max_cache_shape(self) -> Optional[int]:
        return self.max_cache_len

    # what transformers 5.x asks a cache for when it builds the causal mask of a forward (masking_utils._preprocess_mask_arguments):
    # queries start at the committed length, keys are the rows update() returns ([0, committed + new))
    def get_query_offset(self, layer_idx=0):
        return self.cache_length

    def get_mask_sizes(self, query_length, layer_idx=0):
        return self.cache_length + int(query_length), 0

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        """cache.py:103-115"""
        n = key_states.shape[2]
        k_out, v_out = self.key_cache[layer_idx], self.value_cache[layer_idx]
        k_out.narrow(2, self.cache_length, n).copy_(key_states)
        v_out.narrow(2, self.cache_length, n).copy_(value_states)
        if layer_idx == 0:
            self.last_length = self.cache_length + n
        return k_out.narrow(2, 0, self.last_length), v_out.narrow(2, 0, self.last_length)

    def select_indices(self, indices: Optional[torch.Tensor] = None, accept_length: int = 1):
        """cache.py:118-133"""
        start = self.cache_length
        if indices is not None and accept_length > 0:
            idx = indices.reshape(-1).to(device=self.storage.device, dtype=torch.int32).contiguous()
            samd_hip.check(samd_hip.lib().samd_kv_compact_indices_vt(
                samd_hip._ptr(self._ptrs), 2 * self.num_layers, self.num_layers if self.v_transposed else 0, self.num_key_value_heads, self.max_cache_len,
                self.head_dim, self.storage.element_size(), start, samd_hip._ptr(idx), int(accept_length), samd_hip.current_stream()))
        self.cache_length += int(accept_length)


class SamdCache(SamdStaticCache):
    """cache_type="dynamic" (cache.py:8-34: HF DynamicCache + crop).  Growing and cropping a per-layer tensor list is
    the same observable behaviour as keeping the accepted rows in pre-allocated storage, so the dynamic flavour maps
    onto the static one here."""

    def __init__(self, num_hidden_layers=None, config=None, max_cache_len=None, device=None, dtype=torch.float16):
        if config is None:
            raise samd_hip.SamdError("SamdCache needs the LM config to size its storage")
        super().__init__(config, batch_size=1, max_cache_len=max_cache_len, device=device, dtype=dtype)

    def crop(self, length):
        self.cache_length = min(self.cache_length, length)
