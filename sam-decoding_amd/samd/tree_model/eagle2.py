"""EAGLE-2 draft head plugin (reference: samd/tree_model/eagle2/eagle2.py:12-70, eagle2_model.py:583-975).

Not built yet in this round: the registry entry exists so that SamdConfig(tree_method="eagle2") resolves, and
construction fails loudly instead of silently drafting nothing.  See DESIGN.md (scope table, row 18).
"""
from .tree import TreeModel


class Eagle2(TreeModel):

    def __init__(self, config, lm, dtype, device) -> None:
        super().__init__()
        raise NotImplementedError("the EAGLE-2 draft head is not implemented in this build (DESIGN.md, scope row 18)")
