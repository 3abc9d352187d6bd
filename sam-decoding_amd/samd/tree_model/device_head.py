"""The EAGLE draft head on the library's own kernels (gfx950): `DeviceHead` runs Eagle2Head's decoder layer through
samd_hip.llama.LlamaRunner in draft-head mode -- the weight-streaming GEMM, RoPE + K/V write, the tree-mask attention kernel
and the SiLU epilogue -- instead of a few dozen eager PyTorch launches per forward, and shares the base model's lm_head
(packed copy included).  The head keeps its own KV cache; `L` (its length) lives on the device like the base model's.

EAGLE-2's tree levels are STATEFUL, as in the reference (eagle2_model.py:856-913): level i runs its 8 new rows only, writes their
K/V behind the rows of the earlier levels ([L + 8 i, L + 8 i + 8) of the head's cache) and attends to the L accepted tokens plus
the ancestors among the 8 (i + 1) tree rows -- the attention kernel's visible-prefix mask (samd_attention_block: keys < L visible to
every row, key L + j by bit j).  Nothing has to be rolled back afterwards: L only ever counts accepted tokens, the next draft
overwrites the tree rows.  Every head forward (the accepted tokens' extension, then one 8-row forward per level) replays its row
bucket's hipGraph; the tree logic in between stays a handful of device-side PyTorch ops (`eagle2_draft`).  EAGLE v1's static tree keeps the older STATELESS form (`tree`): the forward of level i
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
        self._levels_graph = None                         # EAGLE-2: (hipGraph of the five levels + re-rank, tokens, parents)
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

    def _level_eager(self, i, ids, hidden, anc_rows):
        """level() without the per-forward graph: plain launches (inside the whole-draft capture)"""
        n = ids.numel()
        torch.add(self.L, n * i, out=self.Lw)
        self.n.fill_(n)
        self.x_buf[:n].copy_(self._x(ids, hidden))
        self.mask_buf.copy_(_mask_rows(anc_rows))
        b = self.runner.forward_rows(8, self.tok, self.level_pos[i], self.mask_buf, self.Lw, self.n, x_in=self.x_buf, d_vis=self.L)
        return b["x"][:n], b["logits"][:n]

    def eagle2_draft(self, head, hidden_states, input_ids):
        """Eagle2Head.topk_generate on the library's kernels: hidden_states [T, H] of the accepted tokens, input_ids [T + 1] ->
        (tokens [63], parents [63]).  The accepted tokens enter the head's cache through the per-bucket forward graphs (`extend`);
        the five levels and the top-62 re-rank -- five 8-row forwards and ~150 small PyTorch ops with fixed shapes -- replay as ONE
        hipGraph (`_levels_graph`): launched one by one they keep the host busy for longer (3.7 ms) than the GPU needs (2 ms).
        Only ONE graph of this kind exists per head: graphs that contain PyTorch allocations stopped replaying correctly once a third
        one was captured (ROCm 7.2; the second one faults, scripts/_debug_eagle2.py) -- so the extension, whose row bucket varies,
        stays outside it."""
        last_hidden, last_logits = self.extend(hidden_states, input_ids[1:])
        if head.trace is not None or os.environ.get("SAMD_EAGLE_GRAPH", "1") == "0":      # decision traces copy to the host: eager
            return head._expand_levels(self, last_hidden, last_logits, input_ids[-1:].clone())
        if self._levels_graph is None:
            self.lv_hidden = torch.zeros_like(last_hidden)
            self.lv_logits = torch.zeros_like(last_logits)
            self.lv_sample = torch.zeros(1, dtype=torch.long, device=last_hidden.device)
            level, self.level = self.level, self._level_eager
            try:
                head._expand_levels(self, self.lv_hidden, self.lv_logits, self.lv_sample)       # warm-up: tree rows land beyond L
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = head._expand_levels(self, self.lv_hidden, self.lv_logits, self.lv_sample)
            finally:
                self.level = level
            self._levels_graph = (g, out[0], out[1])
        self.lv_hidden.copy_(last_hidden)
        self.lv_logits.copy_(last_logits)
        self.lv_sample.copy_(input_ids[-1:])
        self._levels_graph[0].replay()
        return self._levels_graph[1].clone(), self._levels_graph[2].clone()

    def expand(self, key, fn, *inputs):
        """run `fn(*inputs) -> tuple of tensors`, a whole tree expansion (fixed shapes, data-dependent values, no host round
        trip).  The head forwards inside it replay their per-bucket hipGraphs.  (Capturing the WHOLE expansion as one graph was
        tried twice: with the library GEMM of fc inside the capture it faulted with an illegal address at the 7B
        configuration; with every GEMM on the streaming kernel it replays correctly and changes nothing -- 5.37 vs 5.37 ms per
        EAGLE step, 6.79 vs 6.82 for EAGLE-2: the expansion is bound by its six head forwards, not by the host.)"""
        return tuple(fn(*inputs))
