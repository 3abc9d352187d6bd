"""Suffix automata of the SAM-only variant (occurrence counts + top-8 successors), backed by libsamd_hip."""
from .dyn_sam import DynSAM
from .static_sam import StaticSAM
from .utils import build_sam, dump_sam, load_sam
