"""SamdModel -- speculative decoding with suffix-automaton drafts, SAM-only variant.

Same constructor, methods and return values as samd_sam_only/samd_model.py:25-333 of the reference.  Two execution
forms share one device state:

  * generate() / stream_generate(): the fused form.  Per step ONE hipGraph replay runs verify forward -> row arg-max ->
    {greedy posterior, SAM update, next lookup, draft, tree buffers} -> KV compaction, and the host reads back a
    704-byte report to apply the reference's stopping rules (samd_model.py:216-235).
  * prefill() / decode() / update_state(): the granular form with the reference's intermediate tensors
    (sample_p, Candidates, best_candidate, accept_length), for tools that drive the steps themselves.

`lm` may be a transformers LlamaForCausalLM (its weights are walked by samd_hip.llama.LlamaRunner), a LlamaRunner, or
any verifier object with prefill/verify/compact/bucket (samd_hip.engine.ScriptedVerifier in tests).
"""
import os
from collections import namedtuple
from typing import Dict, Optional, Union

import torch
import torch.nn as nn

import samd_hip
from samd_hip.engine import DecodeEngine
from samd_hip.llama import LlamaRunner
from .cache import SamdCache, SamdStaticCache
from .draft import DraftModel
from .model_patch import patch_dict
from .model_patch.llama import mask_rows_u64
from .samd_config import ForwardState, ForwardType, MaskState, SamdConfig
from .utils import CandidateType, OptionalTensor, SamdGenerationConfig, eval_posterior, eval_posterior_nodes, gen_candidates

Outputs = namedtuple('Outputs', ['output_ids', 'decode_tokens', 'decode_steps', 'accepet_length_per_step'])


def _is_verifier(obj):
    return all(hasattr(obj, a) for a in ("prefill", "verify", "compact", "bucket"))


class SamdModel(nn.Module):

    def __init__(self,
        samd_config: SamdConfig,
        lm,
        draft: DraftModel,
        eos_token_id: int,
        dtype: torch.dtype,
        device: str,
        stop_token_id: Optional[int] = None,
    ) -> None:
        super().__init__()
        self.samd_config = samd_config
        self.gen_config: SamdGenerationConfig = None
        self.eos_token = eos_token_id
        self.stop_token = stop_token_id

        object.__setattr__(self, "lm", lm)          # not registered as a sub-module: the runner owns the HBM copy
        self.draft = draft
        self.dtype = dtype
        self.device = device

        self.base_seq_position_ids: torch.Tensor = None
        self.base_tree_attn_mask: torch.Tensor = None
        self.base_tree_position_ids: torch.Tensor = None
        self.base_tree_retrieve_indices: torch.Tensor = None
        self.seq_position_ids: torch.Tensor = None
        self.tree_attn_mask: torch.Tensor = None
        self.tree_position_ids: torch.Tensor = None
        self.tree_retrieve_indices: torch.Tensor = None

        self.cache: Union[SamdCache, SamdStaticCache] = None
        self.forward_state = ForwardState(None)
        self.mask_state = MaskState(None)
        self.verifier = lm if _is_verifier(lm) else None
        self._runner: LlamaRunner = None                    # built from an HF module on first set_cache, kept across cache sizes
        self.engine: DecodeEngine = None
        self.lookup_stats = {"sequence": [0, 0], "tree": [0, 0]}      # type -> [steps, accepted tokens]

        self.init_buffers()
        self.register_forward_patch()

    # ---- set-up ------------------------------------------------------------------------------------------------------
    def register_forward_patch(self):
        """samd_model.py:65-77.  The reference re-binds HF methods; here the LM's type selects the runner factory that
        replaces its forward (built in set_cache once max_cache_len is known)."""
        self._runner_factory = None
        for cls, entries in patch_dict.items():
            if isinstance(self.lm, cls):
                self._runner_factory = dict(entries)["forward"]
        if self.verifier is None and self._runner_factory is None:
            raise samd_hip.SamdError(f"SamdModel: unsupported lm type {type(self.lm).__name__} "
                                     "(expected transformers.LlamaForCausalLM, samd_hip.llama.LlamaRunner or a verifier)")

    def init_seq_position_ids(self):
        return torch.arange(0, self.samd_config.max_predicts, dtype=torch.long, device=self.device).unsqueeze(0)

    def init_buffers(self):
        self.base_seq_position_ids = self.init_seq_position_ids()

    def update_buffers(self, buffers_kwargs: Dict[str, Optional[torch.Tensor]]):
        """samd_model.py:89-94"""
        self.seq_position_ids = buffers_kwargs.get("seq_position_ids", self.base_seq_position_ids)
        self.tree_attn_mask = buffers_kwargs.get("tree_attn_mask", self.base_tree_attn_mask)
        self.tree_position_ids = buffers_kwargs.get("tree_position_ids", self.base_tree_position_ids)
        self.tree_retrieve_indices = buffers_kwargs.get("tree_retrieve_indices", self.base_tree_retrieve_indices)
        self.mask_state.set_state(self.tree_attn_mask)

    def _lm_config(self):
        return getattr(self.lm, "config", None)

    def set_cache(self, generation_config: SamdGenerationConfig):
        """samd_model.py:176-191: allocate the static cache on first use, reset it afterwards; also sizes the dynamic
        automaton's arena (prompt + generated tokens <= max_cache_len) and builds the runner / engine."""
        max_len = generation_config.max_cache_len
        if self.verifier is None:
            cfg = self._lm_config()
            if self.cache is None or self.cache.max_cache_len != max_len:
                print("init static cache...")
                cls = SamdCache if self.samd_config.cache_type == "dynamic" else SamdStaticCache
                kw = dict(config=cfg, max_cache_len=max_len, device=self.device, dtype=self.dtype)
                self.cache = cls(**kw) if cls is SamdCache else cls(cfg, batch_size=1, max_cache_len=max_len, device=self.device,
                                                                    dtype=self.dtype, hf_device_map=getattr(self.lm, "hf_device_map", None))
                if self._runner is None:
                    self._runner = self._runner_factory(self.lm, max_len, self.dtype, self.device, kv=self.cache.storage)
                    if os.environ.get("SAMD_RELEASE_HF_WEIGHTS", "0") == "1":
                        self.lm.to("cpu")               # the runner owns its own copies (row-major + packed); opt-in because
                        torch.cuda.empty_cache()        # callers may keep using the HF module on the GPU (tests do)
                else:
                    self._runner.resize_cache(max_len, self.cache.storage)     # same weights; only KV / rope tables change
                self.cache.set_v_transposed(self._runner.v_transposed)     # attention mode "block" keeps V^T in the same storage
                self.verifier = self._runner
                self.engine = None
            else:
                self.cache.reset()
        else:
            # a ready-made runner / verifier was built for ITS max_len: K/V rows past it would be dropped and keys clamped, so a
            # longer generation_config is an error here, not a silent truncation
            r = self.verifier if isinstance(self.verifier, LlamaRunner) else getattr(self.verifier, "runner", None)
            limit = getattr(r, "max_len", None)
            if limit is not None and max_len > limit:
                raise samd_hip.SamdError(f"generation_config.max_cache_len {max_len} exceeds the runner's max_cache_len {limit}")
            if isinstance(self.verifier, LlamaRunner) and self.cache is None:
                self.cache = _RunnerCacheView(self.verifier)
        session = self.draft.ensure_capacity(max_len + samd_hip.MAX_DRAFT)
        # the verifier says how wide a draft it can run; the session's parameters are clamped to that here, with a warning, not mid-generation
        from .sam._common import clamp_to_verifier
        if hasattr(self.draft, "tree_model"):               # samd (full variant): the automata draft n_predicts-token sequences
            asked, what = getattr(self.samd_config, "n_predicts", 0), "n_predicts"
        else:
            asked, what = getattr(self.samd_config, "max_predicts", 0), "max_predicts"
        cap_before = getattr(self.draft, "draft_cap", None)
        if clamp_to_verifier(self.draft, self.verifier, int(asked or 0), what) != cap_before and cap_before is not None:
            self.engine = None                               # parameters changed: the engine holds a copy
        if self.engine is None or self.engine.session is not session:
            self.engine = self._make_engine(session)

    def _make_engine(self, session):
        return DecodeEngine(self.verifier, session, self.draft.static_automaton(), self.draft.params())

    # ---- granular form -------------------------------------------------------------------------------------------------
    def prefill(self, input_ids: torch.Tensor, attention_mask: torch.Tensor = None):
        """samd_model.py:96-114 -> sample_p [1, V]"""
        self.forward_state.forward_type = ForwardType.prefill
        session = self.draft.session()
        last_logits = self.verifier.prefill(session, input_ids.reshape(-1))
        self.draft.update(tokens=input_ids.squeeze(0))
        n = input_ids.shape[-1]
        if self.cache is not None:
            self.cache.last_length = n
            self.cache.set_length()
        if last_logits is None:
            raise samd_hip.SamdError("this verifier does not expose logits; use generate()")
        logits = last_logits.reshape(1, -1)
        return logits if self.gen_config.greedy else torch.softmax(logits.float(), dim=-1)

    def decode(self, sample_p: torch.Tensor, length: int):
        """samd_model.py:116-156 -> (sample_p, new_tokens)"""
        candidates = gen_candidates(sample_p, self.base_tree_retrieve_indices, self.draft, self.samd_config, self.gen_config,
                                    self.device)
        self.update_buffers(candidates.buffers_kwargs)
        n = candidates.tokens.shape[-1]
        if candidates.type == CandidateType.sequence:
            self.forward_state.forward_type = ForwardType.seq_decode
            rel = self.seq_position_ids[0, :n]
            mask_rows = self.verifier.pf_mask if hasattr(self.verifier, "pf_mask") else None
        else:
            self.forward_state.forward_type = ForwardType.tree_decode
            rel = self.tree_position_ids[0]
            mask_rows = mask_rows_u64(self.tree_attn_mask)
        tree_logits = self.verifier.forward_tokens(self.draft.session(), candidates.tokens[0], rel, mask_rows, n, length).unsqueeze(0)
        # (the reference gathers candidate_logits = tree_logits[retrieve] here, samd_model.py:140-146; eval_posterior_nodes reads the
        # per-node rows through the retrieve table instead, so sampling warps <= 64 rows and no [leaves, depth, V] tensor exists)
        is_tree = candidates.type != CandidateType.sequence
        candidate_indices = OptionalTensor(self.tree_retrieve_indices if is_tree else None)
        best_candidate, accept_length, sample_p = eval_posterior_nodes(tree_logits.squeeze(0), self.tree_retrieve_indices if is_tree else None,
                                                                       candidates.candidate_tokens, self.gen_config)
        new_tokens = self.update_state(best_candidate, accept_length, candidates.candidate_tokens, candidate_indices)
        self.lookup_stats[candidates.type.value][0] += 1
        self.lookup_stats[candidates.type.value][1] += len(new_tokens)
        return sample_p, new_tokens

    def update_state(self, best_candidate: torch.Tensor, accept_length: torch.Tensor, candiate_tokens: torch.Tensor,
                     candidate_indices: OptionalTensor):
        """samd_model.py:158-174"""
        tokens = candiate_tokens[best_candidate][:accept_length]
        indices = candidate_indices.apply(lambda x: x[best_candidate][:accept_length]).data
        self.draft.update(tokens=tokens)
        a = int(accept_length.item())
        if self.cache is not None:
            self.cache.select_indices(indices, a)
            self.draft.session().set_cache_length(self.cache.cache_length)
        return tokens.tolist()

    # ---- fused form ------------------------------------------------------------------------------------------------------
    def _truncate(self, new_ids):
        """samd_model.py:220-226: cut at EOS, else at the stop token (inclusive)."""
        if self.eos_token in new_ids:
            return new_ids[:new_ids.index(self.eos_token) + 1], True
        if self.stop_token is not None and self.stop_token in new_ids:
            return new_ids[:new_ids.index(self.stop_token) + 1], True
        return new_ids, False

    def _run(self, input_ids, generation_config, max_steps):
        """generator over decode steps: yields (new_ids, StepReport) after each step."""
        if generation_config is None:
            generation_config = SamdGenerationConfig()
        self.gen_config = generation_config
        assert input_ids.shape[0] == 1, "Only support batch_size == 1"  # [1, N]
        sampled = not generation_config.greedy
        if sampled:
            self.set_cache(generation_config)
            # the fused sampling step (engine.step_sampled: one host synchronisation per step) needs the verify forward's logits on
            # the device and the plain SAM-only engine; anything else decodes through the granular prefill() / decode() loop
            has_logits = isinstance(self.verifier, LlamaRunner) or hasattr(self.verifier, "runner") or bool(getattr(self.verifier, "with_logits", False))
            fused = (os.environ.get("SAMD_FUSED_SAMPLING", "1") != "0" and has_logits
                     and getattr(self.engine, "supports_fused_sampling", lambda: False)())
            if not fused:
                yield from self._run_granular(input_ids, generation_config, max_steps)
                return
        self.set_cache(generation_config)
        rep = self.engine.start_sampled(input_ids) if sampled else self.engine.start(input_ids)
        input_length = input_ids.shape[-1]
        decode_tokens = 0
        try:
            for _ in range(max_steps):
                if input_length + decode_tokens + self.samd_config.max_predicts >= generation_config.max_cache_len:
                    break
                kind = "sequence" if rep.type == 0 else "tree"
                rep = self.engine.step_sampled(rep, generation_config) if sampled else self.engine.step(rep.n)
                if rep.error:
                    raise RuntimeError(f"dynamic suffix automaton capacity exceeded (status {rep.error})")
                new_ids, stop = self._truncate(rep.tokens)
                decode_tokens += len(new_ids)
                self.lookup_stats[kind][0] += 1
                self.lookup_stats[kind][1] += len(new_ids)
                if self.cache is not None:
                    self.cache.cache_length = self.cache.last_length = input_length + decode_tokens
                yield new_ids, rep
                if stop or decode_tokens >= generation_config.max_new_tokens:
                    break
        finally:
            if sampled:                       # the start token drawn for a step that will not happen (engine._draw)
                self.engine.undo_pending_draw()

    def _run_granular(self, input_ids, generation_config, max_steps):
        """the reference's own loop over prefill()/decode() (sampling needs the logits on the host side of the API)."""
        self.set_cache(generation_config)
        self.draft.reset()
        if self.cache is not None:
            self.cache.reset()
        sample_p = self.prefill(input_ids, None)
        input_length, decode_tokens = input_ids.shape[-1], 0
        for _ in range(max_steps):
            if input_length + decode_tokens + self.samd_config.max_predicts >= generation_config.max_cache_len:
                break
            sample_p, new_ids = self.decode(sample_p, input_length + decode_tokens)
            new_ids, stop = self._truncate(new_ids)
            decode_tokens += len(new_ids)
            yield new_ids, None
            if stop or decode_tokens >= generation_config.max_new_tokens:
                break

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: torch.Tensor = None,
                 generation_config: SamdGenerationConfig = None) -> Outputs:
        """samd_model.py:193-237"""
        if generation_config is None:
            generation_config = SamdGenerationConfig()
        input_ids_list = input_ids.squeeze(0).tolist()
        input_length = input_ids.shape[-1]
        decode_tokens, decode_steps, accepet_length_per_step = 0, 0, []
        for new_ids, _ in self._run(input_ids, generation_config, generation_config.max_new_tokens):
            input_ids_list.extend(new_ids)
            decode_steps += 1
            decode_tokens += len(new_ids)
            accepet_length_per_step.append(len(new_ids))
        input_ids_list = [input_ids_list[:input_length + generation_config.max_new_tokens]]
        return Outputs(input_ids_list, decode_tokens, decode_steps, accepet_length_per_step)

    @torch.no_grad()
    def stream_generate(self, input_ids: torch.Tensor, tokenizer, generation_config: SamdGenerationConfig = None):
        """samd_model.py:239-285: yields {"text": decoded continuation} after every step."""
        if generation_config is None:
            generation_config = SamdGenerationConfig()
        out = []
        for new_ids, _ in self._run(input_ids, generation_config, generation_config.max_steps):
            out.extend(new_ids)
            yield {"text": tokenizer.decode(out, skip_special_tokens=True, spaces_between_special_tokens=False,
                                            clean_up_tokenization_spaces=True)}

    def stream_generate_baseline(self, input_ids: torch.Tensor, tokenizer, generation_config: SamdGenerationConfig = None):
        """samd_model.py:287-333 (identical loop in the reference; the AR baseline is max_predicts=1)."""
        yield from self.stream_generate(input_ids, tokenizer, generation_config)


class _RunnerCacheView:
    """cache bookkeeping when SamdModel is handed a LlamaRunner that already owns its KV storage."""

    def __init__(self, runner):
        s = runner.shape
        self.storage, self.max_cache_len = runner.kv, runner.max_len
        self.cache_length = self.last_length = 0
        self._ptrs, self._dims = runner.kv_ptrs, (2 * s.layers, s.kv_heads, runner.max_len, s.head_dim, runner.kv.element_size())
        self._n_vt = s.layers if runner.v_transposed else 0

    def reset(self):
        self.cache_length = self.last_length = 0

    def set_length(self):
        self.cache_length = self.last_length

    def get_seq_length(self, layer_idx=0):
        return self.cache_length

    def select_indices(self, indices=None, accept_length=1):
        if indices is not None and accept_length > 0:
            idx = indices.reshape(-1).to(device=self.storage.device, dtype=torch.int32).contiguous()
            n_t, hk, ml, hd, eb = self._dims
            samd_hip.check(samd_hip.lib().samd_kv_compact_indices_vt(samd_hip._ptr(self._ptrs), n_t, self._n_vt, hk, ml, hd, eb, self.cache_length,
                                                                     samd_hip._ptr(idx), int(accept_length), samd_hip.current_stream()))
        self.cache_length += int(accept_length)
