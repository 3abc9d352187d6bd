"""EAGLE (v1) draft head plugin with a STATIC draft tree (reference: samd/tree_model/eagle/eagle.py:13-75,
eagle_model.py:783-845, eagle_utils.py:75-140, utils.py:62-212).

The head is the one EAGLE-2 uses (one Llama decoder layer without input layer-norm over fc([embed(token_{t+1}) ;
hidden_t]), logits from the base model's lm_head) -- `Eagle2Head.forward` is shared.  What differs is the tree: a fixed
list of "choices" (paths of top-k ranks, samd/config/eagle.json), expanded level by level:

  row 0            = the top-k of the last accepted position
  rows of level d  = the top-k of every depth-d node that has children, in sorted-path order
  draft token of a node with path p = flat[1 + top_k * row(parent(p)) + p[-1]],  flat = [start] + all rows concatenated

`StaticDraftTree` derives everything from the choices once: the parent array (the verify tree's buffers come from the same
wavefront kernel as every other draft), the flat indices, and per level the input selection / hidden-state repeat counts /
attention mask.  One quirk of the reference is kept because it decides which tokens are drafted: inside a level the
reference numbers the parents by order of appearance among the nodes that have children (eagle_utils.py:103-118), not by
their row in the previous level, so a parent whose children are all leaves shifts the selection of the parents after it
(the shipped tree has no such parent; tests/golden/eagle.npz pins both cases).
"""
import json
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .eagle2 import Eagle2, Eagle2Head
from .tree import TreeModel

TOPK = 4          # ranks per node of the sparse tree (eagle/utils.py:6)


class StaticDraftTree:
    """everything the static tree needs, as plain lists (device tensors are made by `to`)."""

    def __init__(self, choices: Sequence[Sequence[int]], top_k: int = TOPK):
        self.top_k = top_k
        paths = sorted((tuple(p) for p in choices), key=lambda p: (len(p), p))
        node = {(): 0}
        node.update({p: i + 1 for i, p in enumerate(paths)})
        self.paths = paths
        self.n = len(paths) + 1
        self.parents: List[int] = [-1] + [node[p[:-1]] for p in paths]
        self.depth: List[int] = [0] + [len(p) for p in paths]
        # rows of the flat candidate array: one per parent, in order of first appearance over the sorted paths
        row: Dict[tuple, int] = {}
        for p in paths:
            row.setdefault(p[:-1], len(row))
        self.flat_index: List[int] = [0] + [1 + top_k * row[p[:-1]] + p[-1] for p in paths]
        # nodes with children (root excluded), sorted; their index in this list is their column in the head's tree mask
        with_children = [p for p in paths if p in row]
        col = {p: i for i, p in enumerate(with_children)}
        self.levels = []
        max_depth = max((len(p) for p in paths), default=0)
        seen = 0
        for d in range(1, max_depth):
            nodes = [p for p in with_children if len(p) == d]
            if not nodes:
                break
            groups: List[tuple] = []                     # parents in order of appearance inside this level
            select, repeat = [], []
            for p in nodes:
                if not groups or groups[-1] != p[:-1]:
                    groups.append(p[:-1])
                    repeat.append(0)
                repeat[-1] += 1
                select.append(p[-1] + top_k * (len(groups) - 1))
            seen += len(nodes)
            mask = [[0] * seen for _ in nodes]
            for r, p in enumerate(nodes):
                for k in range(1, len(p) + 1):           # every ancestor below the root, and the node itself
                    mask[r][col[p[:k]]] = 1
            self.levels.append(dict(select=select, repeat=repeat, mask=mask))

    def reference_buffers(self, device) -> Dict[str, torch.Tensor]:
        """mask / positions / retrieve rows / flat indices in the reference's layout (eagle/utils.py:62-212): retrieve rows
        are the root->leaf paths sorted as index sequences (padding sorts last), padded with -1."""
        n = self.n
        mask = torch.eye(n)
        mask[:, 0] = 1
        for i in range(1, n):
            j = self.parents[i]
            while j > 0:
                mask[i, j] = 1
                j = self.parents[j]
        has_child = set(self.parents)
        rows = []
        for leaf in range(1, n):
            if leaf in has_child:
                continue
            path, j = [], leaf
            while j != -1:
                path.append(j)
                j = self.parents[j]
            rows.append(path[::-1])
        width = max((len(r) for r in rows), default=1)
        big = n + 5
        rows.sort(key=lambda r: r + [big] * (width - len(r)))
        retrieve = torch.tensor([r + [-1] * (width - len(r)) for r in rows] or [[0]], dtype=torch.long)
        return {"tree_attn_mask": mask[None, None].to(device), "tree_indices": torch.tensor(self.flat_index, dtype=torch.long, device=device),
                "tree_position_ids": torch.tensor(self.depth, dtype=torch.long, device=device)[None],
                "tree_retrieve_indices": retrieve.to(device)}


class EagleHead(Eagle2Head):
    """EagleModel (eagle_model.py:576-845): Eagle2Head's forward + the static-tree expansion."""

    def set_tree(self, tree: StaticDraftTree):
        dev = self.device
        self.tree = tree
        self._levels = [dict(select=torch.tensor(l["select"], dtype=torch.long, device=dev),
                             repeat=torch.tensor(l["repeat"], dtype=torch.long, device=dev),
                             mask=torch.tensor(l["mask"], dtype=torch.float32, device=dev)) for l in tree.levels]
        # ancestor-or-self matrices over the nodes-with-children of levels 0..i (for the stateless device forward)
        self._anc, seen = [], 0
        for l in self._levels:
            n_l, width = l["mask"].shape
            full = torch.zeros((width, width), device=dev)
            if self._anc:
                full[:seen, :seen] = self._anc[-1]
            full[seen:width] = l["mask"]
            self._anc.append(full)
            seen = width

    @torch.no_grad()
    def topk_generate(self, hidden_states, input_ids, head_weight):
        """hidden_states [T, H] of the accepted tokens, input_ids [T+1] (accepted tokens + the sampled one), head_weight
        [V, H] -> the rows of top-k token ids, concatenated [n_rows, top_k] (eagle_model.py:783-845, greedy branch)."""
        k, dev = self.tree.top_k, hidden_states.device
        out, kv = self.forward(hidden_states, input_ids[1:], past=self.stable_kv)
        self.stable_kv = kv
        pos_len = kv[0].shape[1]
        prev = out[-1:]
        logits = F.linear(prev, head_weight)
        rows, past = [], kv
        for lvl in self._levels:
            top = self._topk(logits, k).indices
            rows.append(top)
            ids = top.reshape(-1)[lvl["select"]]
            hidden_in = torch.repeat_interleave(prev[:lvl["repeat"].numel()], lvl["repeat"], dim=0)
            prev, past = self.forward(hidden_in, ids, past=past, position_ids=torch.full((ids.numel(),), pos_len, device=dev),
                                      tree_mask=lvl["mask"])
            pos_len += 1
            logits = F.linear(prev, head_weight)
        rows.append(self._topk(logits, k).indices)
        return torch.cat(rows, dim=0)


    @torch.no_grad()
    def topk_generate_device(self, dh, hidden_states, input_ids, head_weight=None):
        """topk_generate with every head forward on the library's kernels (device_head.DeviceHead); same selections.  A
        level's forward carries the nodes-with-children of all levels so far (stateless levels, see device_head.py)."""
        prev, logits = dh.extend(hidden_states, input_ids[1:])
        return dh.expand("eagle", lambda h, lg: (self._expand_device(dh, h, lg),), prev, logits)[0]

    def _expand_device(self, dh, prev, logits):
        """the static levels on device tensors only (fixed shapes, no host round trip): capturable"""
        k, dev = self.tree.top_k, prev.device
        rows = []
        x_rows = torch.empty((0, self.hidden), dtype=prev.dtype, device=dev)
        depth = torch.empty(0, dtype=torch.int32, device=dev)
        for i, lvl in enumerate(self._levels):
            top = self._topk(logits, k).indices
            rows.append(top)
            ids = top.reshape(-1)[lvl["select"]]
            hidden_in = torch.repeat_interleave(prev[:lvl["repeat"].numel()], lvl["repeat"], dim=0, output_size=int(ids.numel()))
            n0 = x_rows.shape[0]
            x_rows = torch.cat((x_rows, dh._x(ids, hidden_in)), dim=0)
            depth = torch.cat((depth, torch.full((ids.numel(),), i, dtype=torch.int32, device=dev)))
            out_all, logits_all = dh.tree(x_rows, depth, self._anc[i])
            prev, logits = out_all[n0:].clone(), logits_all[n0:].clone()
        rows.append(self._topk(logits, k).indices)
        return torch.cat(rows, dim=0)


class Eagle(Eagle2):
    """TreeModel plugin over EagleHead (reference wrapper: eagle/eagle.py:13-75).  State handling (update / reset) is
    EAGLE-2's; the draft is the static tree filled from the flat candidate rows."""
    fused = False

    def __init__(self, config, lm, dtype: torch.dtype, device: str, head: Optional[EagleHead] = None) -> None:
        TreeModel.__init__(self)
        self.dtype, self.device = dtype, device
        self.lm_head = self._find_lm_head(lm)
        if head is None:
            tc = dict(config.tree_config or {})
            head = EagleHead(tc, dtype=dtype, device=device, bias=tc.get("bias", True))
            path = os.path.join(config.tree_model_path or "", "pytorch_model.bin")
            if not os.path.exists(path):
                raise FileNotFoundError(f"EAGLE weights not found: {path}")
            head.load_state(torch.load(path, map_location="cpu"))
        self.model = head
        tree = getattr(config, "tree", None) if config is not None else None
        if getattr(head, "tree", None) is None:
            if tree is None:
                raise ValueError("Eagle needs the static tree choices (SamdConfig.tree)")
            head.set_tree(StaticDraftTree(tree))
        self.tree = head.tree
        self._parents = torch.tensor(self.tree.parents, dtype=torch.long, device=device)
        self._flat = torch.tensor(self.tree.flat_index, dtype=torch.long, device=device)
        self.accept_tokens = self.accept_hidden_states = None
        self.device_head = self._make_device_head(lm)

    def gen_draft_device(self, start_token: torch.Tensor):
        """-> (tokens, parents) on the device; consumes the accumulated state (eagle.py:55-69)."""
        ids = torch.cat((self.accept_tokens.to(torch.long), start_token.reshape(1).to(torch.long)), dim=-1)
        hs = self.accept_hidden_states.to(self.model.dtype)
        self.accept_tokens = self.accept_hidden_states = None
        if self.device_head is not None:
            rows = self.model.topk_generate_device(self.device_head, hs, ids)
        else:
            rows = self.model.topk_generate(hs, ids, self.lm_head.to(self.model.dtype))
        flat = torch.cat((start_token.reshape(1).to(torch.long), rows.reshape(-1)))
        return flat[self._flat], self._parents

    def gen_draft_from_step(self, hidden_rows, views, n_accepted, n_rows):
        """EAGLE-2's device-to-device step path builds EAGLE-2's DYNAMIC tree; the static tree has no such form: always the general
        path (update() + gen_draft_device())."""
        return None

    def gen_draft(self, start_token: int):
        """eagle.py:55-69 -> (tokens, {}): the static buffers of gen_buffers() apply."""
        st = torch.tensor([start_token], dtype=torch.long, device=self.device)
        tokens, self.last_parents = self.gen_draft_device(st)      # the granular decode installs the draft from these
        return tokens.tolist(), {}

    def gen_buffers(self):
        """eagle.py:71-75"""
        buf = self.tree.reference_buffers(self.device)
        self.tree_indices = buf["tree_indices"]
        return buf


def load_tree_choices(path: str) -> List[List[int]]:
    with open(path, "r") as f:
        return json.load(f)["tree_choices"]
