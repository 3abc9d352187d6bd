"""SamdConfig of the full variant: SAM sequence drafts + an auxiliary tree-draft model.

Mirrors samd/samd_config.py:9-43 of the reference (same fields, defaults and __post_init__ loading rules).  The default
Token-Recycle tree (61 nodes; level sizes 1/7/20/21/8/4, the shape of the reference's config/token_recycle.json) is
kept here as child lists; `tree_path` still loads a JSON {"tree_adj": {"0": [...], ...}} file.
"""
import json
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Literal, Optional

from samd_sam_only.samd_config import ForwardState, ForwardType, MaskState  # noqa: F401  same carriers in both variants

# node -> children (rank-ordered: child k of a node receives the parent's k-th most likely successor)
TOKEN_RECYCLE_TREE: List[List[int]] = [
    [1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13], [14, 15, 16, 17, 18], [19, 20, 21], [22, 23], [24, 25], [26], [27],
    [28, 29, 30], [31, 32], [33, 34], [35], [36], [], [37, 38, 39], [40, 41], [42], [43], [], [44], [45], [], [46], [], [47],
    [], [48], [], [49, 50], [51], [], [52], [], [], [], [], [], [53, 54], [], [], [55], [], [], [], [56], [], [], [], [],
    [57, 58], [], [59], [], [60], [], [], [], [], [], [], []]


def load_token_recycle(tree_path: Optional[str] = None):
    """samd_config.py:67-80"""
    if tree_path is None:
        return [list(c) for c in TOKEN_RECYCLE_TREE]
    if not os.path.isabs(tree_path):
        tree_path = os.path.join(os.path.dirname(__file__), "config", tree_path)
    with open(tree_path, "r") as f:
        tree_adj = json.load(f)["tree_adj"]
    return [tree_adj[str(i)] for i in range(len(tree_adj))]


# the static EAGLE (v1) draft tree the reference ships (config/eagle.json): paths of top-4 ranks, 25 nodes below the root
EAGLE_TREE_CHOICES: List[List[int]] = [
    [0], [1], [2], [3], [0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0],
    [0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [0, 1, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0],
    [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1],
]


def load_eagle(tree_model_path: str, tree_path: Optional[str] = None):
    """samd_config.py:83-91: static EAGLE tree choices (default: the shipped tree; `tree_path` = a JSON with
    'tree_choices', relative paths resolved against this package's config/) + the draft head's config.json"""
    if tree_path is None:
        tree = [list(c) for c in EAGLE_TREE_CHOICES]
    else:
        if not os.path.isabs(tree_path):
            tree_path = os.path.join(os.path.dirname(__file__), "config", tree_path)
        with open(tree_path, "r") as f:
            tree = json.load(f)["tree_choices"]
    with open(os.path.join(tree_model_path, "config.json")) as f:
        tree_config = json.load(f)
    return tree, tree_config


def load_eagle2(tree_model_path: str):
    """samd_config.py:94-96"""
    with open(os.path.join(tree_model_path, "config.json")) as f:
        return json.load(f)


@dataclass
class SamdConfig:
    n_predicts: int = field(default=40)
    max_predicts: int = field(default=70)
    len_threshold: int = field(default=5)
    len_bias: int = field(default=5)
    cache_type: Literal["dynamic", "static"] = field(default="static")
    use_last_hidden_states: bool = field(default=False)
    tree_method: Literal["token_recycle", "eagle", "eagle2"] = field(default="token_recycle")
    tree_model_path: Optional[str] = field(default=None)
    tree_path: Optional[str] = field(default=None)
    tree: Optional[List[List[int]]] = field(default=None)
    tree_config: Optional[Dict[str, Any]] = field(default=None)

    def __post_init__(self):
        from samd_hip import MAX_DRAFT
        if self.n_predicts < 1:
            raise ValueError("n_predicts must be >= 1")
        if self.n_predicts > MAX_DRAFT:
            # as samd_sam_only.SamdConfig: drafts are capped at MAX_DRAFT nodes (lossless)
            import warnings
            warnings.warn(f"n_predicts = {self.n_predicts}: drafts are capped at {MAX_DRAFT} nodes on this implementation "
                          "(output tokens are unaffected)", RuntimeWarning, stacklevel=2)
        if self.tree is None:
            if self.tree_method == "token_recycle":
                self.tree = load_token_recycle(self.tree_path)
            elif self.tree_method == "eagle":
                if self.tree_config is None:
                    self.tree, self.tree_config = load_eagle(self.tree_model_path, self.tree_path)
                else:
                    self.tree = [list(c) for c in EAGLE_TREE_CHOICES]
                self.use_last_hidden_states = True
            elif self.tree_method == "eagle2":
                if self.tree_config is None:
                    self.tree_config = load_eagle2(self.tree_model_path)
                self.use_last_hidden_states = True
            else:
                raise ValueError
