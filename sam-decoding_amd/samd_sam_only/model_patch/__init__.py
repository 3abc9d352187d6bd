"""Patch registry (reference: samd_sam_only/model_patch/__init__.py:1-7).

The reference registers monkey patches for HF's LlamaModel._update_causal_mask / LlamaForCausalLM.forward.  Here the
"patch" of a LlamaForCausalLM is its replacement by a LlamaRunner that walks the module's weights with the gfx950
kernels; the tables keep the reference's shape {module type: [(name, fn)]} so SamdModel.register_forward_patch reads
the same way.
"""
from .llama import patch_dict, attn_patch_dict, tree_attention, tree_decode_mask
