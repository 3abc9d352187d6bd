"""Candidate construction and posterior evaluation in their host-visible (granular) form.

Reference semantics: samd_sam_only/utils.py:66-104 (gen_candidates) and :107-184 (eval_posterior).  SamdModel.generate()
does not come through here -- the fused step kernel applies the same rules on the device (sam_device.h: do_accept) --
these functions serve SamdModel.decode() and tools that drive single steps.  Given identical logits the greedy branch
is integer-exact with the reference, including its padding quirks (pad token 0 for candidates, the LAST tree node's
logits for -1 retrieve entries)."""
import random
from typing import Callable, Optional

import torch

import samd_hip
from .draft import Candidates, CandidateType


class OptionalTensor:
    """a tensor or nothing, with map semantics (utils.py:19-28)."""

    def __init__(self, data: Optional[torch.Tensor] = None):
        self.data = data

    def apply(self, fn: Callable) -> 'OptionalTensor':
        return self if self.data is None else OptionalTensor(fn(self.data))


def device_argmax(logits: torch.Tensor) -> torch.Tensor:
    """arg-max over the last dimension by the library kernel (first maximum wins, like torch.argmax) -> int64."""
    vocab = logits.shape[-1]
    rows = logits.reshape(-1, vocab)
    if not rows.is_cuda or rows.stride(-1) != 1 or rows.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise samd_hip.SamdError("device_argmax needs contiguous CUDA logits in f16/bf16/f32 (there is no CPU path)")
    out = torch.zeros(rows.shape[0], dtype=torch.int32, device=rows.device)
    if rows.shape[0]:
        samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(rows), samd_hip.torch_dtype_code(rows.dtype), rows.shape[0], vocab,
                                                       rows.stride(0), None, samd_hip._ptr(out), samd_hip.current_stream()))
    return out.to(torch.long).reshape(logits.shape[:-1])


def gen_candidates(sample_p, tree_retrieve_indices, draft, samd_config, gen_config, device):
    """start token (arg-max, or a multinomial draw when sampling) -> draft.lookup -> Candidates.  Tree drafts are gathered
    through the retrieve table after appending the pad token 0, so that -1 entries select it."""
    start = device_argmax(sample_p).reshape(-1)[0] if gen_config.greedy else torch.multinomial(sample_p, 1).reshape(-1)[0]
    kind, draft_tokens, buffers = draft.lookup(int(start.item()))
    as_row = torch.tensor([draft_tokens], dtype=torch.long, device=device)
    if kind == CandidateType.sequence:
        return Candidates(kind, as_row, as_row, buffers)
    retrieve = buffers.get("tree_retrieve_indices", tree_retrieve_indices)
    padded = torch.cat((as_row[0], torch.zeros(1, dtype=torch.long, device=device)))
    return Candidates(kind, as_row, padded[retrieve], buffers)


def _greedy(logits, candidates):
    """longest prefix of every candidate that the arg-max chain reproduces; the first longest candidate wins."""
    predicted = device_argmax(logits)[:, :-1]
    agree = (candidates[:, 1:] == predicted).to(torch.int32)
    accepted = torch.cumprod(agree, dim=1).sum(dim=1)                 # per candidate
    longest = accepted.max()
    best = torch.argmax(accepted).to(torch.long) if longest != 0 else torch.zeros((), dtype=torch.long, device=candidates.device)
    return best, longest + 1, logits[best, longest].view(1, -1)


def _sampled(logits, candidates, config):
    """utils.py:142-184: walk the candidate trie depth by depth; at each depth try the distinct next tokens of the surviving
    candidates in row order, accepting token x with probability p(x) under the warped distribution (host `random`), and
    renormalising the residual after every rejection."""
    n_rows, depth = candidates.shape
    prefix, n_acc, best = candidates[0][:1], 1, 0
    residual, rejected_last = None, False
    while n_acc < depth:
        rejected_last = False
        alive = (candidates[:, :n_acc] == prefix).all(dim=1)
        anchor = int(torch.nonzero(alive, as_tuple=True)[0][0])
        residual = torch.softmax(config.logits_processor(None, logits[anchor, n_acc - 1][None])[0], dim=0)
        seen, grown = set(), False
        for row in range(n_rows):
            token = int(candidates[row, n_acc]) if bool(alive[row]) else -1
            if token == -1 or token in seen:
                continue
            seen.add(token)
            if random.random() <= float(residual[token]):
                prefix = torch.cat((prefix, candidates[row, n_acc][None]))
                n_acc, best, grown = n_acc + 1, row, True
                break
            residual[token] = 0
            residual = residual / residual.sum()
            rejected_last = True
        if not grown:
            break
    if rejected_last and n_acc != depth:
        sample_p = residual
    else:
        sample_p = torch.softmax(logits[best, n_acc - 1], dim=0)
    dev = candidates.device
    return torch.tensor(best, dtype=torch.long, device=dev), torch.tensor(n_acc, dtype=torch.long, device=dev), sample_p.view(1, -1)


def _sampled_nodes(node_logits, retrieve, candidates, config):
    """_sampled_device with ONE warped row per draft node: node_logits [n, V] are the verify forward's rows, retrieve [C, D] (or None
    for a sequence draft: cell (0, j) = node j) says which node a candidate cell stands on (-1 = the last node, samd_model.py:144).
    HF's warpers (a top-p sort per row) and the softmax then run over n <= 64 rows instead of C * D gathered ones, and no
    [C, D, V] tensor is materialised."""
    import numpy as np
    n, V = node_logits.shape
    C_, D = candidates.shape
    probs = torch.softmax(config.logits_processor(None, node_logits), dim=-1).contiguous()
    rowmap = (retrieve.to(torch.int32) if retrieve is not None else torch.arange(D, dtype=torch.int32, device=node_logits.device).view(1, D)).contiguous()
    cand = candidates.to(torch.long).contiguous()
    n_u = C_ * D + 8
    state = random.getstate()
    u = torch.from_numpy(np.asarray([random.random() for _ in range(n_u)], dtype=np.float64)).to(node_logits.device)
    work = torch.empty(V, dtype=probs.dtype, device=node_logits.device)
    out = torch.zeros(5, dtype=torch.int32, device=node_logits.device)
    samd_hip.check(samd_hip.lib().samd_posterior_sampled_nodes(samd_hip._ptr(probs), samd_hip.torch_dtype_code(probs.dtype), samd_hip._ptr(rowmap), n,
                                                               samd_hip._ptr(cand), C_, D, V, samd_hip._ptr(u), n_u, samd_hip._ptr(work), samd_hip._ptr(out),
                                                               samd_hip.current_stream()))
    best, n_acc, used, residual, status = out.tolist()
    random.setstate(state)
    for _ in range(used):
        random.random()
    if status:
        raise samd_hip.SamdError("samd_posterior_sampled ran out of uniforms")
    if residual:
        sample_p = work
    else:
        node = int(rowmap[best, n_acc - 1])
        sample_p = torch.softmax(node_logits[node if 0 <= node < n else n - 1], dim=0)
    dev = candidates.device
    return torch.tensor(best, dtype=torch.long, device=dev), torch.tensor(n_acc, dtype=torch.long, device=dev), sample_p.view(1, -1)


def eval_posterior_nodes(node_logits, retrieve, candidates, config):
    """eval_posterior for callers that still hold the verify forward's per-node rows and the retrieve table (SamdModel.decode): the
    gather logits[retrieve] of samd_model.py:144 is not materialised.  Same results as eval_posterior(node_logits[retrieve], ...)."""
    usable = (not config.greedy and node_logits.is_cuda and candidates.shape[1] <= samd_hip.MAX_DRAFT
              and node_logits.dtype in (torch.float16, torch.bfloat16, torch.float32))
    if usable:
        return _sampled_nodes(node_logits, retrieve, candidates, config)
    gathered = node_logits[retrieve] if retrieve is not None else node_logits.unsqueeze(0)
    return eval_posterior(gathered, candidates, config)


def _sampled_device(logits, candidates, config):
    """the same walk in ONE kernel (samd_posterior_sampled) and one host round trip, where the reference issues a device
    synchronisation per examined token (`x.item()`, `r <= acp` on a device scalar): HF's warpers and the softmax run over all
    (row, position) logits at once on the device; the kernel consumes uniforms in the reference's order.  RNG contract: the k-th
    uniform a step examines is the k-th value `random.random()` would have returned -- a block of values is drawn under a saved
    generator state, the kernel reports how many it used, the state is restored and advanced by exactly that many, so the host
    stream stays where the reference's would be."""
    import numpy as np
    C_, D, V = candidates.shape[0], candidates.shape[1], logits.shape[-1]
    rows = logits.reshape(C_ * D, V)
    probs = torch.softmax(config.logits_processor(None, rows), dim=-1).contiguous()
    cand = candidates.to(torch.long).contiguous()
    n_u = C_ * D + 8                                          # one uniform per distinct (depth, token) tried: never more than the cells
    state = random.getstate()
    u = torch.from_numpy(np.asarray([random.random() for _ in range(n_u)], dtype=np.float64)).to(logits.device)
    work = torch.empty(V, dtype=probs.dtype, device=logits.device)
    out = torch.zeros(5, dtype=torch.int32, device=logits.device)
    samd_hip.check(samd_hip.lib().samd_posterior_sampled(samd_hip._ptr(probs), samd_hip.torch_dtype_code(probs.dtype), samd_hip._ptr(cand), C_, D, V,
                                                         samd_hip._ptr(u), n_u, samd_hip._ptr(work), samd_hip._ptr(out), samd_hip.current_stream()))
    best, n_acc, used, residual, status = out.tolist()
    random.setstate(state)
    for _ in range(used):
        random.random()
    if status:
        raise samd_hip.SamdError("samd_posterior_sampled ran out of uniforms")
    sample_p = work if residual else torch.softmax(logits[best, n_acc - 1], dim=0)
    dev = candidates.device
    return torch.tensor(best, dtype=torch.long, device=dev), torch.tensor(n_acc, dtype=torch.long, device=dev), sample_p.view(1, -1)


def eval_posterior(logits: torch.Tensor, candidates: torch.Tensor, config):
    """logits [C, depth, V], candidates [C, depth] -> (best_candidate, accept_length, next sample_p [1, V]).  Sampling on device
    tensors runs in the library's kernel; on host tensors (tools, fixtures) in the plain restatement above."""
    if config.greedy:
        return _greedy(logits, candidates)
    if logits.is_cuda and candidates.shape[1] <= samd_hip.MAX_DRAFT and logits.dtype in (torch.float16, torch.bfloat16, torch.float32):
        return _sampled_device(logits, candidates, config)
    return _sampled(logits, candidates, config)
