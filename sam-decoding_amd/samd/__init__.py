"""samd -- drop-in for the reference package of the same name (samd/__init__.py:1-5): SAM sequence drafts combined
with a tree-draft plugin (Token Recycle; EAGLE-2 pending), backed by libsamd_hip.so.  No CPU fallback."""
from .samd_config import SamdConfig
from .samd_model import SamdModel
from .utils import SamdGenerationConfig
from .sam import build_sam, load_sam, dump_sam
from .draft import DraftModel
