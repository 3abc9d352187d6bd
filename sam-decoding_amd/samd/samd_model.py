"""SamdModel of the full variant (reference: samd/samd_model.py:27-322).

Differences to samd_sam_only.SamdModel, as in the reference: sequence drafts are exactly `n_predicts` long with
positions arange(n_predicts) (:79-84); tree drafts come from the tree-draft plugin with STATIC base buffers built once
(:86-92); every verified token and its logits are fed back to the plugin (Token Recycle learns from rejected branches
too, :203-208), at prefill the whole prompt (:117-122).  In the fused form the plugin's table update and tree fill are
two more kernels inside the step's hipGraph (samd_hip.engine.DecodeEngine).
"""
from typing import Dict, Optional

import torch

import samd_hip
from samd_hip.engine import DecodeEngine, TreeModelEngine
from samd_sam_only.model_patch.llama import mask_rows_u64
from samd_sam_only.samd_model import Outputs, SamdModel as _SoSamdModel  # noqa: F401
from .draft import DraftModel
from .samd_config import ForwardType, SamdConfig
from .utils import CandidateType, OptionalTensor, eval_posterior, gen_candidates


class SamdModel(_SoSamdModel):

    def __init__(self, samd_config: SamdConfig, lm, draft: DraftModel, eos_token_id: int, dtype: torch.dtype, device: str,
                 stop_token_id: Optional[int] = None) -> None:
        super().__init__(samd_config, lm, draft, eos_token_id, dtype, device, stop_token_id)

    def init_seq_position_ids(self):
        return torch.arange(0, self.samd_config.n_predicts, dtype=torch.long, device=self.device).unsqueeze(0)

    def init_buffers(self):
        """samd_model.py:86-92"""
        self.seq_position_ids = self.init_seq_position_ids()
        self.base_seq_position_ids = self.seq_position_ids
        buffers = self.draft.tree_model.gen_buffers()       # all None for dynamic-tree plugins (EAGLE-2)
        self.base_tree_attn_mask = buffers["tree_attn_mask"]
        self.base_tree_position_ids = buffers["tree_position_ids"]
        self.base_tree_retrieve_indices = buffers["tree_retrieve_indices"]
        self.mask_state.set_state(self.base_tree_attn_mask)

    def update_buffers(self, buffers_kwargs: Dict[str, Optional[torch.Tensor]]):
        self.tree_attn_mask = buffers_kwargs.get("tree_attn_mask", self.base_tree_attn_mask)
        self.tree_position_ids = buffers_kwargs.get("tree_position_ids", self.base_tree_position_ids)
        self.tree_retrieve_indices = buffers_kwargs.get("tree_retrieve_indices", self.base_tree_retrieve_indices)
        self.mask_state.set_state(self.tree_attn_mask)

    def _make_engine(self, session):
        tm = self.draft.tree_model
        if not getattr(tm, "fused", False):
            return TreeModelEngine(self.verifier, session, self.draft.static_automaton(), self.draft.params(), tm)
        return DecodeEngine(self.verifier, session, self.draft.static_automaton(), self.draft.params(), recycle=tm.table(),
                            recycle_parent=tm.parents)

    # ---- granular form ---------------------------------------------------------------------------------------------------
    def prefill(self, input_ids: torch.Tensor, attention_mask: torch.Tensor = None):
        """samd_model.py:101-128"""
        self.forward_state.forward_type = ForwardType.prefill
        session = self.draft.session()
        tm = self.draft.tree_model

        use_hidden = self.samd_config.use_last_hidden_states

        def on_chunk(tokens, logits, n, hidden):
            tm.update(tokens=tokens[:n].to(torch.long), last_hidden_states=hidden[:n].clone() if use_hidden else None,
                      tree_tokens=tokens[:n], tree_logits=logits[:n])
        last_logits = self.verifier.prefill(session, input_ids.reshape(-1), on_chunk)
        t = input_ids.reshape(-1).to(device="cuda", dtype=torch.int32)
        session.add_tokens(t)
        session.static_walk(self.draft.static_automaton(), t, t.numel(), commit=True)
        n = input_ids.shape[-1]
        if self.cache is not None:
            self.cache.last_length = n
            self.cache.set_length()
        if last_logits is None:
            raise samd_hip.SamdError("this verifier does not expose logits; use generate()")
        logits = last_logits.reshape(1, -1)
        return logits if self.gen_config.greedy else torch.softmax(logits.float(), dim=-1)

    def decode(self, sample_p: torch.Tensor, length: int):
        """samd_model.py:131-182"""
        candidates = gen_candidates(sample_p, self.base_tree_retrieve_indices, self.draft, self.samd_config, self.gen_config,
                                    self.device)
        self.update_buffers(candidates.buffers_kwargs)
        n = candidates.tokens.shape[-1]
        session = self.draft.session()
        if candidates.type == CandidateType.sequence:
            self.forward_state.forward_type = ForwardType.seq_decode
            rel = self.seq_position_ids[0, :n]
            mask_rows = self.verifier.pf_mask if hasattr(self.verifier, "pf_mask") else None
        else:
            self.forward_state.forward_type = ForwardType.tree_decode
            rel = self.tree_position_ids[0]
            mask_rows = mask_rows_u64(self.tree_attn_mask)
            # the tree plugin's draft is not in the session yet (lookup returned host lists): install it for the verifier
            tm = self.draft.tree_model
            static_tree = hasattr(tm, "parents")
            par = torch.tensor(tm.parents, dtype=torch.int32, device="cuda") if static_tree else tm.last_parents.to(torch.int32)
            session.set_draft(candidates.tokens[0].to(torch.int32), par, n, type_=1, reverse=static_tree)
        input_ids = candidates.tokens
        use_hidden = self.samd_config.use_last_hidden_states
        fw = self.verifier.forward_tokens(session, input_ids[0], rel, mask_rows, n, length, **({"return_hidden": True} if use_hidden else {}))
        tree_logits, tree_hidden = (fw[0].unsqueeze(0), OptionalTensor(fw[1])) if use_hidden else (fw.unsqueeze(0), OptionalTensor(None))
        if candidates.type == CandidateType.sequence:
            candidate_logits = tree_logits
            candidate_hidden = tree_hidden.apply(lambda x: x.unsqueeze(0))
            candidate_indices = OptionalTensor(None)
        else:
            candidate_logits = tree_logits.squeeze(0)[self.tree_retrieve_indices]
            candidate_hidden = tree_hidden.apply(lambda x: x[self.tree_retrieve_indices])
            candidate_indices = OptionalTensor(self.tree_retrieve_indices)
        if self.gen_config.greedy:
            best_candidate, accept_length, sample_p = eval_posterior(candidate_logits, candidates.candidate_tokens, self.gen_config)
        else:                                   # sampling: warp the <= 64 node rows, read them through the retrieve table
            from samd_sam_only.posterior import eval_posterior_nodes
            best_candidate, accept_length, sample_p = eval_posterior_nodes(tree_logits.squeeze(0), candidate_indices.data,
                                                                           candidates.candidate_tokens, self.gen_config)
        new_tokens = self.update_state(input_ids.squeeze(0), tree_logits.squeeze(0), best_candidate, accept_length,
                                       candidates.candidate_tokens, candidate_indices, candidate_hidden)
        self.lookup_stats[candidates.type.value][0] += 1
        self.lookup_stats[candidates.type.value][1] += len(new_tokens)
        return sample_p, new_tokens

    def update_state(self, tree_tokens: torch.Tensor, tree_logits: torch.Tensor, best_candidate: torch.Tensor,
                     accept_length: torch.Tensor, candiate_tokens: torch.Tensor, candidate_indices: OptionalTensor,
                     candidate_last_hidden_states: OptionalTensor):
        """samd_model.py:185-211"""
        tokens = candiate_tokens[best_candidate][:accept_length]
        indices = candidate_indices.apply(lambda x: x[best_candidate][:accept_length]).data
        last_hidden_states = candidate_last_hidden_states.apply(lambda x: x[best_candidate][:accept_length]).data
        self.draft.update(tokens=tokens, last_hidden_states=last_hidden_states, tree_tokens=tree_tokens, tree_logits=tree_logits)
        a = int(accept_length.item())
        if self.cache is not None:
            self.cache.select_indices(indices, a)
            self.draft.session().set_cache_length(self.cache.cache_length)
        return tokens.tolist()
