"""EAGLE-2 draft head plugin (reference: samd/tree_model/eagle2/eagle2.py:12-70, eagle2_model.py:583-975).

The head is one Llama decoder layer (without input layer-norm) over fc([embed(token_{t+1}) ; hidden_t]); its logits come
from the base model's lm_head.  `topk_generate` grows a depth-5 / width-8 tree by cumulative log-probability and keeps
the 62 best nodes (eagle2_model.py:820-975).  Differences to the reference, none of them numerical:

  * the whole expansion stays on the device: no `.tolist()` -- the reference walks Python lists to build the mask, the
    position ids and the retrieve table (:915-946); here the expansion returns the PARENT ARRAY and the buffers come from
    the same wavefront kernel that serves every other draft (samd_tree_buffers / samd_session_set_draft);
  * the head's own forward is plain PyTorch-ROCm (a handful of tiny GEMMs and one masked softmax per level); the
    decoder arithmetic follows eagle2_model.py:321-446, :510-581, :704-814.

Weights: `pytorch_model.bin` / `config.json` of an EAGLE-2 checkpoint (config.tree_model_path), or random-init
(tests).  The class is device-agnostic, so its tree logic is checked on CPU against fixtures recorded from the
reference (tests/golden/eagle2.npz).
"""
import math
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .tree import TreeModel


def _rope(x, cos, sin):
    """x [H, T, D]; cos/sin [T, D] (rotate_half convention, eagle2_model.py:99-114)."""
    d = x.shape[-1] // 2
    rot = torch.cat((-x[..., d:], x[..., :d]), dim=-1)
    return x * cos[None] + rot * sin[None]


class Eagle2Head(torch.nn.Module):
    """Eagle2Model (eagle2_model.py:583-975) restated as explicit tensors."""

    def __init__(self, cfg: dict, dtype=torch.float32, device="cpu", bias=True, total_tokens=63, depth=5, top_k=8):
        super().__init__()
        self.hidden = int(cfg["hidden_size"])
        self.heads = int(cfg["num_attention_heads"])
        self.kv_heads = int(cfg.get("num_key_value_heads") or self.heads)
        self.head_dim = self.hidden // self.heads
        self.inter = int(cfg["intermediate_size"])
        self.vocab = int(cfg["vocab_size"])
        self.eps = float(cfg.get("rms_norm_eps", 1e-6))
        self.theta = float(cfg.get("rope_theta", 10000.0))
        self.top_k, self.depth, self.total_tokens = top_k, depth, total_tokens - 1
        self.dtype, self.device = dtype, torch.device(device)
        mk = lambda *s: torch.nn.Parameter(torch.zeros(s, dtype=dtype, device=self.device), requires_grad=False)
        kvd = self.kv_heads * self.head_dim
        self.embed_tokens = mk(self.vocab, self.hidden)
        self.fc_w, self.fc_b = mk(self.hidden, 2 * self.hidden), (mk(self.hidden) if bias else None)
        self.q, self.k, self.v, self.o = mk(self.hidden, self.hidden), mk(kvd, self.hidden), mk(kvd, self.hidden), mk(self.hidden, self.hidden)
        self.gate, self.up, self.down = mk(self.inter, self.hidden), mk(self.inter, self.hidden), mk(self.hidden, self.inter)
        self.post_ln = mk(self.hidden)
        self.stable_kv = None
        inv = 1.0 / (self.theta ** (torch.arange(0, self.head_dim, 2, dtype=torch.float32, device=self.device) / self.head_dim))
        self._inv_freq = inv

    # checkpoint names of the reference's module tree (eagle2_model.py:612-637)
    _NAMES = {"embed_tokens.weight": "embed_tokens", "fc.weight": "fc_w", "fc.bias": "fc_b",
              "layers.0.self_attn.q_proj.weight": "q", "layers.0.self_attn.k_proj.weight": "k",
              "layers.0.self_attn.v_proj.weight": "v", "layers.0.self_attn.o_proj.weight": "o",
              "layers.0.mlp.gate_proj.weight": "gate", "layers.0.mlp.up_proj.weight": "up", "layers.0.mlp.down_proj.weight": "down",
              "layers.0.post_attention_layernorm.weight": "post_ln"}

    def load_state(self, state: Dict[str, torch.Tensor]):
        for name, attr in self._NAMES.items():
            if name in state and getattr(self, attr) is not None:
                getattr(self, attr).data.copy_(torch.as_tensor(state[name]).to(device=self.device, dtype=self.dtype))

    def random_init(self, seed=0, std=0.05):
        g = torch.Generator(device="cpu").manual_seed(seed)
        for attr in set(self._NAMES.values()):
            p = getattr(self, attr)
            if p is None:
                continue
            if attr == "post_ln":
                p.data.fill_(1.0)
            else:
                p.data.copy_((torch.randn(p.shape, generator=g) * std).to(self.dtype))

    def reset(self):
        self.stable_kv = None

    trace = None          # set to a list to record every top-k decision of an expansion as (values, indices): parity tests follow

    def _topk(self, x, k):
        """torch.topk(x, k, dim=-1); the reference's expansion is a sequence of exactly these calls (eagle2_model.py:848-905)"""
        top = torch.topk(x, k, dim=-1)
        if self.trace is not None:
            self.trace.append((top.values.detach().float().cpu(), top.indices.detach().cpu()))
        return top

    # ---- one forward of the head (eagle2_model.py:704-814) --------------------------------------------------------------
    def forward(self, hidden_states, input_ids, past=None, position_ids=None, tree_mask=None):
        """hidden_states [T, H], input_ids [T] -> (out [T, H], (K, V) incl. past, each [H_kv, L+T, D])."""
        T = hidden_states.shape[0]
        L = 0 if past is None else past[0].shape[1]
        if position_ids is None:
            position_ids = torch.arange(L, L + T, device=hidden_states.device)
        emb = self.embed_tokens[input_ids].to(hidden_states.dtype)
        h = F.linear(torch.cat((emb, hidden_states), dim=-1), self.fc_w, self.fc_b)
        # additive mask [T, L+T]: causal over the new block, everything visible in the past; the tree mask overrides the
        # trailing block (eagle2_model.py:693-700)
        mask = torch.zeros((T, L + T), dtype=torch.float32, device=h.device)
        if T > 1:
            mask[:, L:] = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min, device=h.device), diagonal=1)
        if tree_mask is not None:
            t0, t1 = tree_mask.shape
            blk = mask[-t0:, -t1:]
            blk[tree_mask == 0] = torch.finfo(torch.float32).min
        # decoder layer 0: no input layer-norm (eagle2_model.py:516-519)
        q = F.linear(h, self.q).view(T, self.heads, self.head_dim).transpose(0, 1)
        k = F.linear(h, self.k).view(T, self.kv_heads, self.head_dim).transpose(0, 1)
        v = F.linear(h, self.v).view(T, self.kv_heads, self.head_dim).transpose(0, 1)
        ang = position_ids.to(torch.float32)[:, None] * self._inv_freq[None, :]
        ang = torch.cat((ang, ang), dim=-1)
        cos, sin = ang.cos().to(h.dtype), ang.sin().to(h.dtype)
        q, k = _rope(q, cos, sin), _rope(k, cos, sin)
        if past is not None:
            k, v = torch.cat((past[0], k), dim=1), torch.cat((past[1], v), dim=1)
        present = (k, v)
        rep = self.heads // self.kv_heads
        kk, vv = k.repeat_interleave(rep, dim=0), v.repeat_interleave(rep, dim=0)
        att = torch.matmul(q, kk.transpose(1, 2)) / math.sqrt(self.head_dim) + mask[None].to(q.dtype)
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        a = torch.matmul(att, vv).transpose(0, 1).reshape(T, self.hidden)
        h = h + F.linear(a, self.o)
        x = h.float()
        x = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps)).to(h.dtype) * self.post_ln
        h = h + F.linear(F.silu(F.linear(x, self.gate)) * F.linear(x, self.up), self.down)
        return h, present

    # ---- tree expansion (eagle2_model.py:820-975) -----------------------------------------------------------------------------
    @torch.no_grad()
    def topk_generate(self, hidden_states, input_ids, head_weight):
        """hidden_states [T, H] of the accepted tokens, input_ids [T+1] (accepted tokens + the sampled one),
        head_weight [V, H] (lm_head).  -> (draft_tokens int64 [n], parents int64 [n], parent -1 for the root); n = 63."""
        top_k, dev = self.top_k, hidden_states.device
        sample_token = input_ids[-1:]
        out, kv = self.forward(hidden_states, input_ids[1:], past=self.stable_kv)
        self.stable_kv = kv
        pos_len = kv[0].shape[1]
        last_hidden = out[-1:]
        logp = torch.log_softmax(F.linear(last_hidden, head_weight).float(), dim=-1)      # fp32: no -inf ties in half precision
        top = self._topk(logp, top_k)
        scores = top.values[0]
        scores_list, parents_list, tokens_list = [scores[None]], [torch.zeros(1, dtype=torch.long, device=dev)], [top.indices]
        ids = top.indices[0]
        in_hidden = last_hidden.repeat(top_k, 1)
        tree_mask = torch.eye(top_k, device=dev)
        cs_index = torch.arange(top_k, device=dev)
        past = kv
        for i in range(self.depth):
            out, past = self.forward(in_hidden, ids, past=past, position_ids=torch.full((top_k,), pos_len, device=dev), tree_mask=tree_mask)
            pos_len += 1
            bias = 1 + top_k ** 2 * max(0, i - 1) + (top_k if i > 0 else 0)
            parents_list.append(cs_index + bias)
            logp = torch.log_softmax(F.linear(out, head_weight).float(), dim=-1)
            top = self._topk(logp, top_k)
            cu = top.values + scores[:, None]
            best = self._topk(cu.view(-1), top_k)
            cs_index, scores = best.indices, best.values
            rows = cs_index // top_k
            in_hidden = out[rows]
            ids = top.indices.reshape(-1)[cs_index]
            tokens_list.append(top.indices)
            scores_list.append(cu)
            tree_mask = torch.cat((tree_mask[rows], torch.eye(top_k, device=dev)), dim=1)
        all_scores = torch.cat(scores_list, dim=0).view(-1)
        all_tokens = torch.cat(tokens_list, dim=0).view(-1)
        keep = torch.sort(self._topk(all_scores, self.total_tokens).indices).values
        draft_tokens = torch.cat((sample_token, all_tokens[keep]), dim=0)
        draft_parents = torch.cat(parents_list, dim=0)[keep // top_k].long()
        mask_index = torch.searchsorted(keep, draft_parents - 1, right=False)
        mask_index[draft_parents == 0] = -1
        parents = torch.cat((torch.full((1,), -1, dtype=torch.long, device=dev), mask_index + 1), dim=0)
        # a kept node whose parent was not kept (possible only on exact score ties) hangs off the root, as the tree-buffer
        # kernel would do anyway
        idx = torch.arange(parents.numel(), device=dev)
        parents = torch.where((parents >= idx) & (idx > 0), torch.zeros_like(parents), parents)
        return draft_tokens, parents


    @torch.no_grad()
    def topk_generate_device(self, dh, hidden_states, input_ids, head_weight=None):
        """topk_generate with every head forward on the library's kernels (device_head.DeviceHead `dh`); same tree logic, stateful
        levels (8 new rows each, earlier levels stay in the head's cache), every forward a hipGraph replay."""
        return dh.eagle2_draft(self, hidden_states, input_ids)

    def _expand_levels(self, dh, last_hidden, last_logits, sample_token):
        """the level loop of topk_generate on device tensors only (fixed shapes, no host round trip): capturable.  Mirrors
        eagle2_model.py:848-913 step by step: the same top-k calls in the same order."""
        top_k, dev = self.top_k, last_hidden.device
        logp = torch.log_softmax(last_logits.float(), dim=-1)
        top = self._topk(logp, top_k)
        scores = top.values[0]
        scores_list, parents_list, tokens_list = [scores[None]], [torch.zeros(1, dtype=torch.long, device=dev)], [top.indices]
        ids = top.indices[0]
        in_hidden = last_hidden.repeat(top_k, 1)
        level_mask = torch.eye(top_k, device=dev)
        cs_index = torch.arange(top_k, device=dev)
        for i in range(self.depth):
            out, logits = dh.level(i, ids, in_hidden, level_mask)
            bias = 1 + top_k ** 2 * max(0, i - 1) + (top_k if i > 0 else 0)
            parents_list.append(cs_index + bias)
            logp = torch.log_softmax(logits.float(), dim=-1)
            top = self._topk(logp, top_k)
            cu = top.values + scores[:, None]
            best = self._topk(cu.view(-1), top_k)
            cs_index, scores = best.indices, best.values
            rows = cs_index // top_k
            in_hidden = out[rows]
            ids = top.indices.reshape(-1)[cs_index]
            tokens_list.append(top.indices)
            scores_list.append(cu)
            level_mask = torch.cat((level_mask[rows], torch.eye(top_k, device=dev)), dim=1)
        all_scores = torch.cat(scores_list, dim=0).view(-1)
        all_tokens = torch.cat(tokens_list, dim=0).view(-1)
        keep = torch.sort(self._topk(all_scores, self.total_tokens).indices).values
        draft_tokens = torch.cat((sample_token.reshape(1), all_tokens[keep]), dim=0)
        draft_parents = torch.cat(parents_list, dim=0)[keep // top_k].long()
        mask_index = torch.searchsorted(keep, draft_parents - 1, right=False)
        mask_index = torch.where(draft_parents == 0, torch.full_like(mask_index, -1), mask_index)
        parents = torch.cat((torch.full((1,), -1, dtype=torch.long, device=dev), mask_index + 1), dim=0)
        idx = torch.arange(parents.numel(), device=dev)
        parents = torch.where((parents >= idx) & (idx > 0), torch.zeros_like(parents), parents)
        return draft_tokens, parents


class Eagle2(TreeModel):
    """TreeModel plugin over Eagle2Head (reference wrapper: eagle2.py:12-70)."""
    fused = False

    def __init__(self, config, lm, dtype: torch.dtype, device: str, head: Optional[Eagle2Head] = None) -> None:
        super().__init__()
        self.dtype, self.device = dtype, device
        self.lm_head = self._find_lm_head(lm)
        if head is None:
            tc = dict(config.tree_config or {})
            head = Eagle2Head(tc, dtype=dtype, device=device, bias=tc.get("bias", True))
            path = os.path.join(config.tree_model_path or "", "pytorch_model.bin")
            if not os.path.exists(path):
                raise FileNotFoundError(f"EAGLE-2 weights not found: {path}")
            head.load_state(torch.load(path, map_location="cpu"))
        self.model = head
        self.accept_tokens: Optional[torch.Tensor] = None
        self.accept_hidden_states: Optional[torch.Tensor] = None
        self.device_head = self._make_device_head(lm)

    def _make_device_head(self, lm):
        """the head on the library's kernels when the base model runs on them too (a samd_hip LlamaRunner, possibly wrapped);
        SAMD_EAGLE_DEVICE_HEAD=0 keeps the PyTorch forward"""
        runner = lm if hasattr(lm, "forward_rows") else getattr(lm, "runner", None)
        if runner is None or not hasattr(runner, "forward_rows") or os.environ.get("SAMD_EAGLE_DEVICE_HEAD", "1") == "0":
            return None
        if self.model.head_dim != 128 or str(self.model.device).startswith("cpu"):
            # a base model on the library's kernels never gets a PyTorch draft head behind its back (no fallback)
            from samd_hip import SamdError
            raise SamdError(f"the draft head of a LlamaRunner base model runs on the gfx950 kernels, which need head_dim 128 and a "
                            f"GPU-resident head (got head_dim {self.model.head_dim} on {self.model.device}); set "
                            f"SAMD_EAGLE_DEVICE_HEAD=0 to ask for the PyTorch head explicitly")
        from .device_head import DeviceHead
        return DeviceHead(self.model, runner)

    @staticmethod
    def _find_lm_head(lm):
        if hasattr(lm, "lm_head"):
            return lm.lm_head.weight
        for holder in (lm, getattr(lm, "runner", None)):
            w = getattr(holder, "w", None)
            if isinstance(w, dict) and "lm_head" in w:
                return w["lm_head"]
        raise ValueError("Eagle2 needs the base model's lm_head weight")

    def reset(self):
        """eagle2.py:34-35"""
        self.model.stable_kv = None
        if self.device_head is not None:
            self.device_head.reset()

    def update(self, tokens: torch.Tensor = None, last_hidden_states: torch.Tensor = None, **kwargs):
        """eagle2.py:37-50: accumulate the accepted tokens and their base-model hidden states until the next draft."""
        tokens = tokens.reshape(-1).to(self.device)
        hs = last_hidden_states.reshape(tokens.numel(), -1).to(self.device)
        self.accept_tokens = tokens if self.accept_tokens is None else torch.cat([self.accept_tokens, tokens], dim=-1)
        self.accept_hidden_states = hs if self.accept_hidden_states is None else torch.cat([self.accept_hidden_states, hs], dim=-2)

    def gen_draft_device(self, start_token: torch.Tensor):
        """-> (tokens, parents) on the device; consumes the accumulated state (eagle2.py:52-63)."""
        ids = torch.cat((self.accept_tokens.to(torch.long), start_token.reshape(1).to(torch.long)), dim=-1)
        hs = self.accept_hidden_states.to(self.model.dtype)
        self.accept_tokens = self.accept_hidden_states = None
        if self.device_head is not None:
            return self.model.topk_generate_device(self.device_head, hs, ids)
        return self.model.topk_generate(hs, ids, self.lm_head.to(self.model.dtype))

    def gen_draft_from_step(self, hidden_rows, views, n_accepted, n_rows):
        """update() + gen_draft_device() for the accepted tokens of one verified step that are still on the device (the engine's
        report block `views`, the verify forward's hidden rows): no host-to-device copies, no PyTorch staging ops.  None when
        the fast path does not apply (accumulated tokens of earlier steps pending, no device head, unusual head shape)."""
        dh = self.device_head
        if dh is None or self.accept_tokens is not None or not dh.fast_step_ok(self.model, n_accepted):
            return None
        return dh.eagle2_draft_step(self.model, hidden_rows, views, n_accepted, n_rows)

    def gen_draft(self, start_token: int):
        """eagle2.py:52-63 -> (tokens, buffers_kwargs); the buffers come from the tree-buffer kernel."""
        from samd_sam_only.sam.static_sam import gen_buffers
        st = torch.tensor([start_token], dtype=torch.long, device=self.device)
        tokens, parents = self.gen_draft_device(st)
        buf = gen_buffers(parents.tolist(), self.device)
        buf["tree_attn_mask"] = buf["tree_attn_mask"].float()
        self.last_parents = parents
        return tokens.view(-1).tolist(), buf

    def gen_buffers(self):
        """eagle2.py:65-70: EAGLE-2 trees are dynamic, there are no static base buffers."""
        return {"tree_attn_mask": None, "tree_position_ids": None, "tree_retrieve_indices": None}
