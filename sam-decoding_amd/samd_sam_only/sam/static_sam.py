"""StaticSAM -- the corpus suffix automaton (occurrence counts, top-8 successors, best-first tree drafts).

Facade over samd_hip.StaticAutomaton (host-built flat image, uploaded to HBM once per GPU) with the method names of
samd_sam_only/sam/static_sam.py.  The cursor lives in a samd_hip.Session (shared with the DraftModel's DynSAM when
bound).  Granular calls synchronise; SamdModel.generate() uses the fused step kernel on the same state.
"""
from dataclasses import dataclass
from typing import Dict, List

import torch

import samd_hip
from ._common import CursorOwner, dev_i32, so_params, tree_buffers_from_draft


def gen_buffers(anc_tree: List[int], device="cuda"):
    """static_sam.py:148-180 on the GPU (samd_tree_buffers): parent array -> mask / depths / retrieve rows."""
    n = len(anc_tree)
    par = dev_i32(anc_tree)
    pos = torch.zeros(n, dtype=torch.int32, device="cuda")
    mask_bool = torch.zeros(n * n, dtype=torch.uint8, device="cuda")
    ret = torch.full((n * n,), -1, dtype=torch.int32, device="cuda")
    shape = torch.zeros(2, dtype=torch.int32, device="cuda")
    samd_hip.check(samd_hip.lib().samd_tree_buffers(samd_hip._ptr(par), n, 0, samd_hip._ptr(pos), None, samd_hip._ptr(mask_bool),
                                                    samd_hip._ptr(ret), samd_hip._ptr(shape), samd_hip.current_stream()))
    nl, md = shape.tolist()
    return {
        "tree_attn_mask": mask_bool.view(1, 1, n, n).bool().to(device),
        "tree_position_ids": pos.to(torch.long).view(1, n).to(device),
        "tree_retrieve_indices": ret[:nl * md].to(torch.long).view(nl, md).to(device),
    }


class StaticSAM(CursorOwner):
    _own_capacity = 8
    KIND = samd_hip.KIND_COUNT

    @dataclass
    class SAMState:                       # static_sam.py:24-29 (also the shape reference pickles carry)
        next: Dict[int, int]
        link: int
        length: int
        cnt_endpos: int

    @staticmethod
    def build(batch_tokens: List[List[int]], eos_token: int, verbose: bool = True):
        """static_sam.py:31-40: all documents into one automaton (+EOS per document), then the top-k tables."""
        sam = StaticSAM()
        sam.add_batch_tokens(batch_tokens, eos_token, verbose)
        sam.init_topk_next()
        return sam

    def __init__(self, max_predicts: int = 40, alpha: float = 4.0, K: int = 8, device: str = "cuda"):
        self.max_predicts = max_predicts
        self.alpha = alpha
        self.K = K
        self.device = device
        self._auto = None                 # samd_hip.StaticAutomaton
        self._pending = []                # documents added but not yet built
        self._eos = None

    # ---- construction ----------------------------------------------------------------------------------------
    def add_batch_tokens(self, batch_tokens: List[List[int]], eos_token: int, verbose: bool = False):
        """static_sam.py:131-135.  Construction is batch-wise here: documents are collected and the automaton is
        built by the native builder on init_topk_next() / first use."""
        if self._auto is not None:
            raise samd_hip.SamdError("StaticSAM is immutable once built (one shared HBM image per GPU)")
        self._pending.extend([list(t) for t in batch_tokens])
        self._eos = eos_token

    def init_topk_next(self):
        """static_sam.py:137-146: the top-8 successor order is part of the 64-byte node layout built natively."""
        if self._auto is None:
            self._auto = samd_hip.StaticAutomaton.build(self._pending, -1 if self._eos is None else self._eos, self.KIND)
            self._pending = []

    @classmethod
    def _from_automaton(cls, auto):
        sam = cls()
        sam._auto = auto
        return sam

    def _automaton(self):
        self.init_topk_next()
        if not self._auto.info()["uploaded"]:
            self._auto.upload()
        return self._auto

    # reference pickles (samd_sam_only/sam/utils.py:20-22) unpickle into THIS class: convert their object graph
    def __setstate__(self, state):
        self.__init__()
        for key in ("max_predicts", "alpha", "K", "device", "n_predicts"):      # utils.py:29-33 copies known attributes
            if key in state and hasattr(self, key):
                setattr(self, key, state[key])
        states = state.get("states")
        if states is not None and len(states) and not isinstance(states, dict):
            link = [s.link for s in states]
            length = [s.length for s in states]
            aux = [getattr(s, "cnt_endpos", getattr(s, "min_endpos", 0)) for s in states]
            deg = [len(s.next) for s in states]
            et = [t for s in states for t in s.next.keys()]
            ed = [d for s in states for d in s.next.values()]
            text = state.get("input_ids") if self.KIND == samd_hip.KIND_ENDPOS else None
            self._auto = samd_hip.StaticAutomaton.from_tables(self.KIND, link, length, aux, deg, et, ed, text)

    def __getstate__(self):
        raise samd_hip.SamdError("use dump_sam(path, sam): the automaton is stored as a flat binary image, not a pickle")

    # ---- state views -------------------------------------------------------------------------------------------
    @property
    def cur_index(self):
        return int(self._sess().export(with_edges=False)["st_index"])

    @property
    def cur_length(self):
        return int(self._sess().export(with_edges=False)["st_length"])

    @property
    def states(self):
        self.init_topk_next()
        e = self._auto.export()
        out, k = [], 0
        for i in range(len(e["link"])):
            d = int(e["deg"][i])
            nxt = dict(zip(e["edge_tok"][k:k + d].tolist(), e["edge_dst"][k:k + d].tolist()))
            k += d
            out.append(self.SAMState(nxt, int(e["link"][i]), int(e["length"][i]), int(e["aux"][i])))
        return out

    @property
    def states_topk_next(self):
        """static_sam.py:140-146 result: per state the first 8 (token, next) pairs in top-k order."""
        self.init_topk_next()
        e = self._auto.export()
        out, k = [], 0
        for d in e["deg"].tolist():
            m = min(d, samd_hip.TOPK)
            out.append(list(zip(e["edge_tok"][k:k + m].tolist(), e["edge_dst"][k:k + m].tolist())))
            k += d
        return out

    # ---- reference API -----------------------------------------------------------------------------------------
    def reset(self):
        """static_sam.py:127-129: only the cursor rewinds."""
        s = self._sess()
        info = s.export(with_edges=False)
        s.set_cursors(int(info["cur_index"]), int(info["cur_length"]), 0, 0)

    def transfer_tokens(self, tokens: List[int]):
        """static_sam.py:118-120"""
        if len(tokens):
            self._sess().static_walk(self._automaton(), dev_i32(tokens), len(tokens), commit=True)

    def transfer_cur_state(self, token: int):
        self.transfer_tokens([token])

    def lookup(self, token: int):
        """static_sam.py:122-125"""
        sess = self._sess()
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        sess.static_walk(self._automaton(), dev_i32([token]), 1, commit=False, d_out=out)
        i, l = out.tolist()
        return i, l

    def transfer_state(self, index: int, length: int, token: int):
        """static_sam.py:98-107 from an arbitrary (index, length): one lane of the batched walk kernel."""
        cur = dev_i32([[index, length]])
        self._automaton().walk(cur, dev_i32([[token]]), commit=True)
        i, l = cur[0].tolist()
        return i, l

    def gen_draft(self, index: int, match_length: int, start_token: int):
        """static_sam.py:182-215 -> (tree tokens, gen_buffers(anc_tree))"""
        s = self._sess()
        s.draft_tree(self._automaton(), so_params(self.max_predicts, self.alpha, self.K), index, match_length, start_token)
        d = s.read_draft()
        return list(d.tokens[:d.n]), tree_buffers_from_draft(d, self.device)

    def gen_buffers(self, anc_tree: List[int]):
        return gen_buffers(anc_tree, self.device)
