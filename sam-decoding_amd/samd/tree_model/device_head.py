"""The EAGLE draft head on the library's own kernels (gfx950): `DeviceHead` runs Eagle2Head's decoder layer through
samd_hip.llama.LlamaRunner in draft-head mode -- the weight-streaming GEMM, RoPE + K/V write, the tree-mask attention kernel
and the SiLU epilogue -- instead of a few dozen eager PyTorch launches per forward, and shares the base model's lm_head
(packed copy included).  The head keeps its own KV cache; `L` (its length) lives on the device like the base model's.

EAGLE-2's tree levels are STATEFUL, as in the reference (eagle2_model.py:856-913): level i runs its 8 new rows only, writes their
K/V behind the rows of the earlier levels ([L + 8 i, L + 8 i + 8) of the head's cache) and attends to the L accepted tokens plus
the ancestors among the 8 (i + 1) tree rows -- the attention kernel's visible-prefix mask (samd_attention_block: keys < L visible to
every row, key L + j by bit j).  Nothing has to be rolled back afterwards: L only ever counts accepted tokens, the next draft
overwrites the tree rows.  Every head forward (the accepted tokens' extension, then one 8-row forward per level) replays its row
bucket's hipGraph; the tree logic in between -- log-softmax + top-k per row, cumulative scores, the level's selection, the next
level's inputs and masks, the final re-rank -- runs in three small kernels (csrc/eagle_kernels.hip; `eagle2_draft`).  EAGLE v1's static tree keeps the older STATELESS form (`tree`): the forward of level i
carries every node-with-children so far, with their ancestor mask.  Positions are the reference's (accepted length + depth).

Arithmetic is the same as Eagle2Head.forward's up to fp16 accumulation order; drafts are verified by the base model either
way, so the generated text cannot change (tests/test_gpu_llama.py checks losslessness through this path)."""
from typing import Optional

import os

import torch
import torch.nn.functional as F

MAX_ROWS = 64


def _mask_rows(anc: torch.Tensor) -> torch.Tensor:
    """anc [n, m] 0/1 (row i sees column j; m <= 64) -> int64[64] bit rows for the attention kernels"""
    n, m = anc.shape
    bits = (anc.to(torch.int64) << torch.arange(m, device=anc.device, dtype=torch.int64)[None, :]).sum(-1)
    out = torch.zeros(MAX_ROWS, dtype=torch.int64, device=anc.device)
    out[:n] = bits
    return out


class DeviceHead:
    def __init__(self, head, base_runner):
        from samd_hip.llama import LlamaRunner, LlamaShape
        dev, dt = base_runner.device, base_runner.dtype
        self.head, self.base = head, base_runner
        cfg = dict(hidden_size=head.hidden, intermediate_size=head.inter, num_hidden_layers=1, num_attention_heads=head.heads,
                   num_key_value_heads=head.kv_heads, head_dim=head.head_dim, vocab_size=base_runner.shape.vocab,
                   max_position_embeddings=base_runner.max_len + MAX_ROWS, rms_norm_eps=head.eps, rope_theta=head.theta)
        ones = torch.ones(head.hidden, dtype=dt, device=dev)
        c = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()
        weights = dict(embed=c(head.embed_tokens), norm=ones, lm_head=base_runner.w["lm_head"],
                       layers=[dict(ln1=ones, wqkv=torch.cat([c(head.q), c(head.k), c(head.v)], 0), wo=c(head.o), ln2=c(head.post_ln),
                                    wgu=torch.cat([c(head.gate), c(head.up)], 0), wdown=c(head.down))])
        self.runner = LlamaRunner(LlamaShape(cfg), weights, base_runner.max_len + MAX_ROWS, dt, dev,
                                  packed_lm_head=base_runner.wp["lm_head"] if base_runner.wp else None, attention="block")
        self.runner.draft_head = True
        self.fc_w, self.fc_b = c(head.fc_w), (c(head.fc_b) if head.fc_b is not None else None)
        # fc([embed ; hidden]) on the streaming GEMM too (K = 2 * hidden): packed weight, zero-padded operand rows, fp32 partials
        import samd_hip
        self._lib, self._dt = samd_hip.lib(), base_runner.dt
        n_fc, k_fc = self.fc_w.shape
        self.fc_packed = None
        if n_fc % 128 == 0 and k_fc % 256 == 0 and k_fc >= 512:
            self.fc_packed = torch.empty_like(self.fc_w)
            samd_hip.check(self._lib.samd_gemm_pack_weights(samd_hip._ptr(self.fc_w), samd_hip._ptr(self.fc_packed), n_fc, k_fc, samd_hip.current_stream()))
            self.fc_in = torch.zeros((MAX_ROWS, k_fc), dtype=dt, device=dev)
            self.fc_part = torch.zeros((8, MAX_ROWS, n_fc), dtype=torch.float32, device=dev)
        self.embed = weights["embed"]
        self.L = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n = torch.zeros(1, dtype=torch.int32, device=dev)
        self.relpos = torch.zeros(MAX_ROWS, dtype=torch.int32, device=dev)
        self.relpos_buf = torch.zeros(MAX_ROWS, dtype=torch.int32, device=dev)
        self.tok = torch.zeros(MAX_ROWS, dtype=torch.int32, device=dev)
        self.x_buf = torch.zeros((MAX_ROWS, head.hidden), dtype=dt, device=dev)
        self.mask_buf = torch.zeros(MAX_ROWS, dtype=torch.int64, device=dev)
        self._graphs = {}                                 # row bucket -> hipGraph of one head forward over the static buffers
        self._step_plans = {}                             # row bucket -> resolved arguments of eagle2_draft_step's launches
        self._e2 = None                                   # EAGLE-2 tree-logic state (samd_e2_state_t), made on first use
        # stateful tree levels (EAGLE-2): write position of the current level, per-level relative positions
        self.Lw = torch.zeros(1, dtype=torch.int32, device=dev)
        self.level_pos = [torch.full((MAX_ROWS,), i, dtype=torch.int32, device=dev) for i in range(8)]
        self.length = 0                                   # host mirror of L: accepted tokens in the head's cache
        chain = [(1 << (i + 1)) - 1 for i in range(MAX_ROWS)]
        self.chain_mask = torch.tensor([r - (1 << 64) if r >= (1 << 63) else r for r in chain], dtype=torch.int64, device=dev)
        self.chain_pos = torch.arange(MAX_ROWS, dtype=torch.int32, device=dev)

    def reset(self):
        self.length = 0
        self.L.zero_()

    def _x(self, ids: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
        """fc([embed(ids) ; hidden]) -> [n, H] (eagle2_model.py:739-741)"""
        n = ids.numel()
        if self.fc_packed is None or n > MAX_ROWS:
            return F.linear(torch.cat((self.embed[ids], hidden.to(self.embed.dtype)), dim=-1), self.fc_w, self.fc_b)
        import samd_hip
        H = self.embed.shape[1]
        rp = 16 if n <= 16 else (32 if n <= 32 else 64)
        self.fc_in[:n, :H] = self.embed[ids]
        self.fc_in[:n, H:] = hidden.to(self.embed.dtype)
        n_fc, k_fc = self.fc_w.shape
        sp = max(2, min(8, self._lib.samd_gemm_splits(n_fc, k_fc, rp), k_fc // 256))
        part = self.fc_part.view(-1)[:sp * rp * n_fc].view(sp, rp, n_fc)            # the kernel's [splits][rows_pad][N] layout
        samd_hip.check(self._lib.samd_gemm_skinny(samd_hip._ptr(self.fc_in), samd_hip._ptr(self.fc_packed), rp, n_fc, k_fc, sp, samd_hip._ptr(part),
                                                  None, self._dt, samd_hip.current_stream()))
        x = part[:, :n].sum(0)
        return (x + self.fc_b.float() if self.fc_b is not None else x).to(self.embed.dtype)

    def _forward(self, x: torch.Tensor, relpos: torch.Tensor, mask: torch.Tensor, level: bool = False):
        """one head forward over n <= 64 rows.  Inputs are staged into fixed buffers and the launch sequence of the row
        bucket is replayed as a hipGraph (captured on first use): ~15 kernel launches become one.  level = a stateful tree level:
        the rows are written at Lw (behind the earlier levels' rows) and see the first L keys plus the tree rows their mask names."""
        n = x.shape[0]
        R = self.runner.bucket(n)
        self.n.fill_(n)
        self.x_buf[:n].copy_(x)
        self.relpos_buf[:MAX_ROWS].copy_(relpos[:MAX_ROWS])
        self.mask_buf.copy_(mask)
        key = ("level", R) if level else R
        g = self._graphs.get(key)
        if level:
            run = lambda: self.runner.forward_rows(R, self.tok, self.relpos_buf, self.mask_buf, self.Lw, self.n, x_in=self.x_buf, d_vis=self.L)
        else:
            run = lambda: self.runner.forward_rows(R, self.tok, self.relpos_buf, self.mask_buf, self.L, self.n, x_in=self.x_buf)
        if os.environ.get("SAMD_HEAD_GRAPH", "1") == "0":          # debugging: plain launches
            run()
            b = self.runner._buffers(R)
            return b["x"][:n], b["logits"][:n]
        if g is None:
            run()                                                  # library handles / workspaces exist before capture
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            self._graphs[key] = g
        g.replay()
        b = self.runner._buffers(R)
        return b["x"][:n], b["logits"][:n]

    def extend(self, hidden_states: torch.Tensor, input_ids: torch.Tensor):
        """the accepted tokens enter the head's cache (causal), 64 rows at a time -> (last output state [1, H], its logits [1, V])"""
        T = hidden_states.shape[0]
        out = None
        for c0 in range(0, T, MAX_ROWS):
            n = min(MAX_ROWS, T - c0)
            out = self._forward(self._x(input_ids[c0:c0 + n], hidden_states[c0:c0 + n]), self.chain_pos, self.chain_mask)
            self.length += n
            self.L.fill_(self.length)
        return out[0][-1:].clone(), out[1][-1:].clone()

    def tree(self, x_rows: torch.Tensor, depth: torch.Tensor, anc: torch.Tensor):
        """all tree rows so far: x_rows [n, H] (fc outputs), depth [n] (0 = children of the last accepted token), anc [n, n]
        ancestor-or-self matrix -> (output states [n, H], logits [n, V]) (views into the runner's buffers: consume before the
        next call)"""
        self.relpos[:depth.numel()] = depth.to(torch.int32)
        return self._forward(x_rows, self.relpos, _mask_rows(anc))

    # ---- EAGLE-2: stateful tree levels -------------------------------------------------------------------------------
    def level(self, i, ids, hidden, anc_rows):
        """tree level i (8 rows) on top of the earlier levels' rows in the cache: ids [8], hidden [8, H], anc_rows [8, 8 (i + 1)] 0/1
        (row r sees tree row j) -> (output states [8, H], logits [8, V]); views into the bucket's buffers: consume before the next call"""
        n = ids.numel()
        torch.add(self.L, n * i, out=self.Lw)
        return self._forward(self._x(ids, hidden), self.level_pos[i], _mask_rows(anc_rows), level=True)

    def _e2_state(self, depth, keep):
        """device arrays of the tree logic (samd_e2_state_t); mask_rows IS the forward's mask buffer"""
        if self._e2 is None or self._e2[0] != (depth, keep):
            import ctypes as C
            import samd_hip
            dev, f32, i32 = self.embed.device, torch.float32, torch.int32
            n_cand = 8 + 64 * depth
            t = dict(row_lse=torch.zeros(8, dtype=f32, device=dev), top_logp=torch.zeros(64, dtype=f32, device=dev), top_idx=torch.zeros(64, dtype=i32, device=dev),
                     scores=torch.zeros(16, dtype=f32, device=dev), cs_index=torch.zeros(8, dtype=i32, device=dev),
                     all_scores=torch.zeros(n_cand, dtype=f32, device=dev), all_tokens=torch.zeros(n_cand, dtype=i32, device=dev),
                     parents_list=torch.zeros(1 + 8 * depth, dtype=i32, device=dev), mask_rows=self.mask_buf,
                     row_src=torch.zeros(8, dtype=i32, device=dev), ids=torch.zeros(8, dtype=i32, device=dev),
                     rec_top_vals=torch.zeros((1 + depth) * 64, dtype=f32, device=dev), rec_top_idx=torch.zeros((1 + depth) * 64, dtype=i32, device=dev),
                     rec_best_vals=torch.zeros(depth * 8, dtype=f32, device=dev), rec_best_idx=torch.zeros(depth * 8, dtype=i32, device=dev),
                     rec_final_vals=torch.zeros(keep, dtype=f32, device=dev), rec_final_idx=torch.zeros(keep, dtype=i32, device=dev))
            st = samd_hip.E2State(**{k: v.data_ptr() for k, v in t.items()})
            out = (torch.zeros(keep + 1, dtype=i32, device=dev), torch.zeros(keep + 1, dtype=i32, device=dev))
            nb = int(self._lib.samd_e2_rowstats_workspace(self.base.shape.vocab))
            self._e2_ws = (torch.zeros(max(nb, 4) // 4, dtype=f32, device=dev), nb)      # split rowstats: partial (m, s) + candidates
            self._e2 = ((depth, keep), t, st, C.byref(st), out)
        return self._e2

    def eagle2_draft(self, head, hidden_states, input_ids):
        """Eagle2Head.topk_generate on the library's kernels: hidden_states [T, H] of the accepted tokens, input_ids [T + 1] ->
        (tokens [keep + 1], parents [keep + 1]).  Per level: the fc projection (streaming GEMM + bias), one 8-row head forward (its
        hipGraph), samd_e2_rowstats (log-softmax + top-8 per row) and samd_e2_select (cumulative scores, top-8 of 64, next level's
        inputs and masks) -- six launches where the PyTorch form issues ~35 small ops and leaves the GPU waiting for the host (3.7 ms
        per draft against 2 ms of GPU work at Llama-3-8B shapes).  Capturing those PyTorch ops in hipGraphs instead faulted on replay
        once several such graphs existed (ROCm 7.2).  Falls back to the PyTorch form when the head's shape
        does not fit the kernels (top_k != 8, depth > 7, fc not streamable)."""
        import samd_hip
        depth, keep = head.depth, head.total_tokens
        if head.top_k != 8 or depth > 7 or self.fc_packed is None or os.environ.get("SAMD_EAGLE_KERNELS", "1") == "0":
            last_hidden, last_logits = self.extend(hidden_states, input_ids[1:])
            return head._expand_levels(self, last_hidden, last_logits, input_ids[-1:].clone())
        last_hidden, last_logits = self.extend(hidden_states, input_ids[1:])
        self._keep_sample = input_ids[-1:].to(torch.long).contiguous()
        tokens, parents = self._e2_levels(head, samd_hip._ptr(last_hidden), samd_hip._ptr(last_logits), self._keep_sample)
        return tokens.to(torch.long), parents.to(torch.long)

    def eagle2_draft_step(self, head, hidden_rows, views, n_accepted, n_rows):
        """eagle2_draft for the accepted tokens of ONE verified step, read where the step left them on the device: hidden_rows
        [R, H] = the verify forward's last hidden states, views = the session's report block (kv_index / acc_tokens / start_token
        pointers), n_accepted = the step's accepted tokens (host copy of the verdict), n_rows = the verified draft's size.
        samd_e2_stage_extend builds the fc input rows, chain mask, positions and n; fc + bias land in the row bucket's own x; the
        extension forward replays its in-place hipGraph: 5 launches + a replay where the general path issues ~25 PyTorch ops and
        two host-to-device copies.  Same arithmetic as extend() + eagle2_draft()."""
        import samd_hip
        T = int(n_accepted)
        R = self.runner.bucket(T)
        plan = self._step_plans.get(R)
        if plan is None:                                       # everything that does not depend on the step, resolved once per row bucket
            L, dt, _ptr = self._lib, self._dt, samd_hip._ptr
            b = self.runner._buffers(R)
            g = self._extend_graph(R)                          # before rows are staged (first use warms up in place)
            if getattr(self, "_sample64", None) is None:
                self._sample64 = torch.zeros(1, dtype=torch.long, device=self.embed.device)
            rp = b["rows_pad"]
            n_fc, k_fc = self.fc_w.shape
            sp = max(2, min(8, L.samd_gemm_splits(n_fc, k_fc, rp), k_fc // 256))
            part = self.fc_part.view(-1)[:sp * rp * n_fc]
            H = self.embed.shape[1]
            plan = self._step_plans[R] = dict(
                g=g, H=H, esz=b["x"].element_size(), x=b["x"].data_ptr(), logits=b["logits"].data_ptr(), lstride=b["logits"].stride(0),
                stage_tail=(_ptr(self.embed), H, self.embed.shape[0], _ptr(self.fc_in), _ptr(self.relpos_buf), _ptr(self.mask_buf), _ptr(self.n), _ptr(self._sample64), dt),
                gemm=(_ptr(self.fc_in), _ptr(self.fc_packed), rp, n_fc, k_fc, sp, _ptr(part), None, dt),
                bias=(_ptr(part), sp, rp * n_fc, _ptr(self.fc_b), _ptr(b["x"])), n_fc=n_fc)
        L, st, check, _ptr = self._lib, samd_hip.current_stream(), samd_hip.check, samd_hip._ptr
        check(L.samd_e2_stage_extend(_ptr(hidden_rows), _ptr(views["kv_index"]), _ptr(views["acc_tokens"]), _ptr(views["start_token"]), T, int(n_rows), *plan["stage_tail"], st))
        check(L.samd_gemm_skinny(*plan["gemm"], st))
        check(L.samd_sum_partials_bias(*plan["bias"], T, plan["n_fc"], self._dt, st))
        plan["g"].replay()
        self.length += T
        H, esz = plan["H"], plan["esz"]
        # no PyTorch op on this path (a tiny one costs the host ~36 us here, scripts/host_launch_cost.py): L += T happens inside the
        # root's select launch, the outputs stay int32 (what the engine installs)
        return self._e2_levels(head, plan["x"] + (T - 1) * H * esz, plan["logits"] + (T - 1) * plan["lstride"] * esz, self._sample64, advance_L=T)

    def fast_step_ok(self, head, n_accepted):
        return (head.top_k == 8 and head.depth <= 7 and self.fc_packed is not None and self.fc_b is not None and 1 <= n_accepted <= MAX_ROWS
                and os.environ.get("SAMD_EAGLE_KERNELS", "1") != "0" and os.environ.get("SAMD_EAGLE_FAST_STEP", "1") != "0")

    def _extend_graph(self, R):
        """hipGraph of one causal extension forward over R rows staged in the bucket's own x (relpos_buf, mask_buf, L, n)"""
        g = self._graphs.get(("extend-in-place", R))
        if g is None:
            b = self.runner._buffers(R)
            run = lambda: self.runner.forward_rows(R, self.tok, self.relpos_buf, self.mask_buf, self.L, self.n, x_in=b["x"])
            self.n.fill_(1)                                    # the warm-up run writes one row at L (overwritten by the real one)
            run()
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            self._graphs[("extend-in-place", R)] = g
        return g

    def _e2_levels(self, head, p_last_hidden, p_last_logits, sample64, advance_L=0):
        """the tree levels + re-rank after the extension: p_last_* = device addresses of the last accepted row's output state / logits;
        advance_L: accepted tokens the extension wrote that L does not count yet.  -> (tokens, parents) int32 [keep + 1], views of
        persistent buffers (consume before the next draft)"""
        import samd_hip
        depth, keep = head.depth, head.total_tokens
        L, st = self._lib, samd_hip.current_stream()
        _, t, _st, st_ref, out = self._e2_state(depth, keep)
        H, V = self.embed.shape[1], self.base.shape.vocab
        n_fc, k_fc = self.fc_w.shape
        sp = max(2, min(8, L.samd_gemm_splits(n_fc, k_fc, 16), k_fc // 256))
        part = self.fc_part.view(-1)[:sp * 16 * n_fc]
        dt = self._dt
        check = samd_hip.check
        ws, ws_bytes = self._e2_ws
        ws_ptr = samd_hip._ptr(ws) if ws_bytes else None
        if ("level-in-place", 8) not in self._graphs:
            torch.add(self.L, advance_L, out=self.Lw)      # a first use warms the graph up: it must write tree rows, not accepted ones,
        level_graph = self._level_graph()                  # and come BEFORE rows are staged (the forward transforms b["x"] in place)
        check(L.samd_e2_rowstats(samd_hip._ptr(p_last_logits), dt, 1, V, V, st_ref, ws_ptr, ws_bytes, st))
        pL, pLw, pn = samd_hip._ptr(self.L), samd_hip._ptr(self.Lw), samd_hip._ptr(self.n)
        check(L.samd_e2_select(st_ref, -1, samd_hip._ptr(p_last_hidden), samd_hip._ptr(self.embed), H, self.embed.shape[0], samd_hip._ptr(self.fc_in),
                               samd_hip._ptr(self.relpos_buf), pL, pLw, pn, int(advance_L), dt, st))      # also L += advance_L, Lw <- L, n <- 8
        b = self.runner._buffers(8)
        for i in range(depth):
            check(L.samd_gemm_skinny(samd_hip._ptr(self.fc_in), samd_hip._ptr(self.fc_packed), 16, n_fc, k_fc, sp, samd_hip._ptr(part), None, dt, st))
            check(L.samd_sum_partials_bias(samd_hip._ptr(part), sp, 16 * n_fc, samd_hip._ptr(self.fc_b), samd_hip._ptr(b["x"]), 8, n_fc, dt, st))
            level_graph.replay()
            check(L.samd_e2_rowstats(samd_hip._ptr(b["logits"]), dt, 8, V, b["logits"].stride(0), st_ref, ws_ptr, ws_bytes, st))
            check(L.samd_e2_select(st_ref, i, samd_hip._ptr(b["x"]), samd_hip._ptr(self.embed), H, self.embed.shape[0], samd_hip._ptr(self.fc_in),
                                   samd_hip._ptr(self.relpos_buf), pL, pLw, pn, 0, dt, st))   # Lw <- L + 8 (i + 1)
        check(L.samd_e2_finish(st_ref, depth, keep, samd_hip._ptr(sample64), samd_hip._ptr(out[0]), samd_hip._ptr(out[1]), st))
        if head.trace is not None:                         # every top-k decision in the reference's order (parity tests)
            torch.cuda.current_stream().synchronize()
            tv, ti = t["rec_top_vals"].view(1 + depth, 8, 8).cpu(), t["rec_top_idx"].view(1 + depth, 8, 8).cpu().long()
            bv, bi = t["rec_best_vals"].view(depth, 8).cpu(), t["rec_best_idx"].view(depth, 8).cpu().long()
            head.trace.append((tv[0, :1], ti[0, :1]))
            for i in range(depth):
                head.trace.append((tv[1 + i], ti[1 + i]))
                head.trace.append((bv[i], bi[i]))
            head.trace.append((t["rec_final_vals"].cpu(), t["rec_final_idx"].cpu().long()))
        return out[0], out[1]

    def _level_graph(self):
        """hipGraph of one 8-row tree-level forward over the fixed buffers (the bucket's x, relpos_buf, mask_buf, Lw, n; visible prefix L)"""
        g = self._graphs.get(("level-in-place", 8))
        if g is None:
            b = self.runner._buffers(8)                           # the level's input rows are staged in the bucket's own x (no copy)
            run = lambda: self.runner.forward_rows(8, self.tok, self.relpos_buf, self.mask_buf, self.Lw, self.n, x_in=b["x"], d_vis=self.L)
            run()
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            self._graphs[("level-in-place", 8)] = g
        return g

    def expand(self, key, fn, *inputs):
        """run `fn(*inputs) -> tuple of tensors`, a whole tree expansion (fixed shapes, data-dependent values, no host round
        trip).  The head forwards inside it replay their per-bucket hipGraphs.  (Capturing the WHOLE expansion as one graph was
        tried twice: with the library GEMM of fc inside the capture it faulted with an illegal address at the 7B
        configuration; with every GEMM on the streaming kernel it replays correctly and changes nothing -- 5.37 vs 5.37 ms per
        EAGLE step, 6.79 vs 6.82 for EAGLE-2: the expansion is bound by its six head forwards, not by the host.)"""
        return tuple(fn(*inputs))
