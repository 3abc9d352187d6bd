"""samd/model_patch of the reference equals samd_sam_only/model_patch modulo comments; one implementation serves both."""
from samd_sam_only.model_patch import attn_patch_dict, patch_dict, tree_attention, tree_decode_mask  # noqa: F401
