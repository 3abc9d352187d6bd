"""SamdGenerationConfig, gen_candidates, eval_posterior (reference: samd_sam_only/utils.py:30-184).

These are the granular, host-visible forms used by SamdModel.decode(); generate() runs the same rules inside the
fused step kernel (samd_session_step).  The greedy branch is integer-exact with the reference given identical
logits (torch.argmax first-maximum tie rule, candidate padding quirks of utils.py:95-96 / samd_model.py:144).
"""
import random
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch

import samd_hip
from .draft import Candidates, CandidateType, DraftModel
from .samd_config import SamdConfig


class OptionalTensor:

    def __init__(self, data: Optional[torch.Tensor] = None):
        self.data = data

    def apply(self, fn: Callable) -> 'OptionalTensor':
        return OptionalTensor(None) if self.data is None else OptionalTensor(fn(self.data))


@dataclass
class SamdGenerationConfig:
    max_steps: int = field(default=512)
    max_new_tokens: int = field(default=512)
    max_cache_len: int = field(default=2048)
    greedy: bool = field(default=True)
    temperature: float = field(default=0.0)
    top_p: float = field(default=0.0)
    top_k: int = field(default=0)
    logits_processor: object = field(default=None)

    def __post_init__(self):
        if not self.greedy:
            assert self.temperature >= 1e-5
            self.logits_processor = self.prepare_logits_processor(self.temperature, self.top_p, self.top_k)

    @staticmethod
    def prepare_logits_processor(temperature: float = 0.0, top_p: float = 0.0, top_k: int = 0):
        """utils.py:50-63: temperature, then nucleus, then top-k warpers (HF LogitsProcessorList)."""
        from transformers.generation.logits_process import (LogitsProcessorList, TemperatureLogitsWarper, TopKLogitsWarper,
                                                            TopPLogitsWarper)
        processors = LogitsProcessorList()
        if temperature >= 1e-5 and temperature != 1.0:
            processors.append(TemperatureLogitsWarper(temperature))
        if 1e-8 <= top_p < 1.0:
            processors.append(TopPLogitsWarper(top_p))
        if top_k > 0:
            processors.append(TopKLogitsWarper(top_k))
        return processors


def device_argmax(logits: torch.Tensor) -> torch.Tensor:
    """torch.argmax(logits, -1) over the last dim with the library kernel (first maximum wins) -> int64 tensor."""
    v = logits.shape[-1]
    flat = logits.reshape(-1, v)
    if not flat.is_cuda or flat.dtype not in (torch.float16, torch.bfloat16, torch.float32) or flat.stride(-1) != 1:
        raise samd_hip.SamdError("device_argmax needs contiguous CUDA logits in f16/bf16/f32 (there is no CPU path)")
    out = torch.zeros(flat.shape[0], dtype=torch.int32, device=flat.device)
    if flat.shape[0] == 0:
        return out.to(torch.long).reshape(logits.shape[:-1])
    samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(flat), samd_hip.torch_dtype_code(flat.dtype), flat.shape[0], v,
                                                   flat.stride(0), None, samd_hip._ptr(out), samd_hip.current_stream()))
    return out.to(torch.long).reshape(logits.shape[:-1])


def gen_candidates(
    sample_p: torch.Tensor,
    tree_retrieve_indices: torch.Tensor,
    draft: DraftModel,
    samd_config: SamdConfig,
    gen_config: SamdGenerationConfig,
    device: torch.device,
):
    """utils.py:66-104: start token -> draft.lookup -> token tensors (tree: candidates gathered through the retrieve
    table with the appended pad token 0)."""
    if gen_config.greedy:
        start_token = int(device_argmax(sample_p).reshape(-1)[0].item())
    else:
        start_token = int(torch.multinomial(sample_p, 1).item())
    candidate_type, tokens, buffers_kwargs = draft.lookup(start_token)
    tree_retrieve_indices = buffers_kwargs.get("tree_retrieve_indices", tree_retrieve_indices)
    if candidate_type == CandidateType.sequence:
        tokens = torch.tensor([tokens], dtype=torch.long, device=device)
        candidate_tokens = tokens
    else:
        padded = torch.tensor(tokens + [0], dtype=torch.long, device=device)
        candidate_tokens = padded[tree_retrieve_indices]
        tokens = torch.tensor([tokens], dtype=torch.long, device=device)
    return Candidates(candidate_type, tokens, candidate_tokens, buffers_kwargs)


def eval_posterior(
    logits: torch.Tensor,
    candidates: torch.Tensor,
    config: SamdGenerationConfig,
):
    """utils.py:107-184.  logits [C, depth, V], candidates [C, depth] -> (best_candidate, accept_length, sample_p)."""
    if config.greedy:
        # utils.py:127-141: longest prefix of each candidate that equals the arg-max chain; first max wins
        hits = (candidates[:, 1:] == device_argmax(logits)[:, :-1]).int()
        per_candidate = torch.cumprod(hits, dim=1).sum(dim=1)
        accept_length = per_candidate.max()
        if accept_length == 0:
            best_candidate = torch.tensor(0, dtype=torch.long, device=candidates.device)
        else:
            best_candidate = torch.argmax(per_candidate).to(torch.long)
        return best_candidate, accept_length + 1, logits[best_candidate, accept_length].view(1, -1)
    # utils.py:142-184: typical-acceptance style sampling over the candidate trie, host RNG (random.random())
    accepted = candidates[0][:1]
    n_acc, best, residual, adjusted = 1, 0, None, False
    for depth in range(1, candidates.shape[1]):
        if depth != n_acc:
            break
        adjusted = False
        alive = (candidates[:, :n_acc] == accepted).all(dim=1)
        first = int(torch.nonzero(alive, as_tuple=True)[0][0])
        row = config.logits_processor(None, logits[first, depth - 1][None])[0]
        residual = torch.softmax(row, dim=0)
        tried = []
        for j in range(candidates.shape[0]):
            if not bool(alive[j]):
                continue
            tok = candidates[j, depth]
            t = int(tok.item())
            if t in tried or t == -1:
                continue
            tried.append(t)
            if random.random() <= float(residual[t]):
                accepted = torch.cat((accepted, tok[None]), dim=0)
                n_acc += 1
                best = j
                break
            residual[t] = 0
            residual = residual / residual.sum()
            adjusted = True
    if adjusted and n_acc != candidates.shape[1]:
        sample_p = residual
    else:
        sample_p = torch.softmax(logits[best, n_acc - 1], dim=0)
    return (torch.tensor(best, dtype=torch.long, device=candidates.device),
            torch.tensor(n_acc, dtype=torch.long, device=candidates.device), sample_p.view(1, -1))
