"""samd_sam_only -- drop-in for the reference package of the same name (samd_sam_only/__init__.py:1-5), backed by
libsamd_hip.so (hand-written gfx950 kernels).  There is no CPU fallback: operations raise without an MI355X."""
from .samd_config import SamdConfig
from .samd_model import SamdModel
from .utils import SamdGenerationConfig
from .sam import build_sam, load_sam, dump_sam
from .draft import DraftModel
