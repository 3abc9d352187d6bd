"""Suffix automata of the full variant (first end positions + corpus text), backed by libsamd_hip."""
from .dyn_sam import DynSAM
from .static_sam import StaticSAM, NullStaticSAM
from .utils import build_sam, dump_sam, load_sam
