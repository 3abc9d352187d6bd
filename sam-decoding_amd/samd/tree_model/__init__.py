"""Tree-draft plugins of the full variant (reference: samd/tree_model/__init__.py:7-14)."""
from typing import Dict

from .tree import TreeModel
from .token_recycle import TokenRecycle
from .eagle2 import Eagle2
from .eagle import Eagle

tree_model_cls: Dict[str, type] = {
    "token_recycle": TokenRecycle,
    "eagle": Eagle,
    "eagle2": Eagle2,
}
