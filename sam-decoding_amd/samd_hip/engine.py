"""DecodeEngine -- one request stream of SAM-Decoding on one MI355X.

What SamdModel.prefill / decode / update_state do between two LM forwards in the reference
(samd_sam_only/samd_model.py:96-174; samd/samd_model.py:101-211) -- arg-max, draft lookup, tree buffers, greedy
posterior, SAM update, KV compaction -- is here a fixed sequence of kernels on one stream with no host round trip:

    verify forward (R rows) -> row arg-max -> [Token-Recycle update] -> k_session{accept, commit, lookup, draft, buffers}
    -> [Token-Recycle draft install] -> KV compaction -> report copy (D2H, pinned)

The sequence is captured once per row bucket R into a hipGraph; the host replays it and reads the 704-byte report
(accepted tokens, next draft size) to apply the reference's stopping rules.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import (F32, MAX_DRAFT, REP_COUNTERS, REP_DMETA, REP_KVINDEX, REP_META, REP_TOKENS, REP_VERDICT, REPORT_INTS,
               Params, SamdError, Session, StaticAutomaton, TokenRecycleTable, require_gpu, torch_dtype_code)


class StepReport:
    """host view of one report block (include/samd_hip.h SAMD_REP_*)."""
    __slots__ = ("type", "n", "n_leaves", "max_depth", "match_dyn", "match_static", "best", "accept", "next_token", "is_tree",
                 "tokens", "kv_index", "steps", "error")

    def __init__(self, r):
        d, v = r[REP_DMETA:REP_DMETA + 16], r[REP_VERDICT:REP_VERDICT + 8]
        self.type, self.n, self.n_leaves, self.max_depth = int(d[0]), int(d[1]), int(d[2]), int(d[3])
        self.match_dyn, self.match_static = int(d[5]), int(d[7])
        self.best, self.accept, self.next_token, self.is_tree = int(v[0]), int(v[1]), int(v[3]), int(v[5])
        self.tokens = r[REP_TOKENS:REP_TOKENS + self.accept].tolist()
        self.kv_index = r[REP_KVINDEX:REP_KVINDEX + self.accept].tolist()
        self.steps = int(r[REP_COUNTERS])
        self.error = int(r[REP_META + 9])


class ScriptedVerifier:
    """Stand-in for the LM forward in tests, smoke() and bench.py's acceptance model: the arg-max of every draft node
    comes from a target stream (samd_scripted_argmax, the device twin of tests/scripted_lm.py).  It exercises every
    non-LM kernel of the step with reproducible accept lengths; it is not a CPU fallback of anything."""

    def __init__(self, target, vocab, device="cuda", with_logits=False):
        require_gpu()
        self.vocab, self.device = int(vocab), torch.device(device)
        self.host_target = [int(t) for t in target]
        self.target = torch.tensor(self.host_target, dtype=torch.int32, device=self.device)
        self.argmax = torch.zeros(MAX_DRAFT, dtype=torch.int32, device=self.device)
        self.with_logits = with_logits
        self.logits = torch.zeros((MAX_DRAFT, self.vocab), dtype=torch.float32, device=self.device) if with_logits else None
        self._base = ((torch.arange(self.vocab, device=self.device, dtype=torch.int64) * 37)) if with_logits else None
        self.dtype = torch.float32

    def next_token(self, ctx):
        k = len(ctx)
        if k < len(self.host_target) and list(ctx) == self.host_target[:k]:
            return self.host_target[k]
        h = 1469598103
        for t in ctx[-3:]:
            h = (h * 1000003 + t + 7) % 2147483647
        return 3 + h % (self.vocab - 3)

    def _rows(self, n_rows):
        """tie-free logits rows whose arg-max is self.argmax (same construction as tests/scripted_lm.py row())."""
        nt = self.argmax[:n_rows].to(torch.int64)
        v = self.vocab
        r = ((self._base[None, :] + 11 * nt[:, None]) % v).to(torch.float32) / float(v)
        idx = torch.arange(n_rows, device=self.device)
        r[idx, nt] += 8.0
        r[idx, (nt * 7 + 1) % v] += 3.0
        self.logits[:n_rows] = r

    def prefill(self, session, input_ids, on_chunk=None):
        ids = [int(t) for t in input_ids.reshape(-1).tolist()]
        n = len(ids)
        session.set_cache_length(n)
        if on_chunk is not None:
            if not self.with_logits:
                raise SamdError("ScriptedVerifier(with_logits=True) is required for Token Recycle")
            for c0 in range(0, n, MAX_DRAFT):
                m = min(MAX_DRAFT, n - c0)
                am = [self.next_token(ids[:c0 + i + 1]) for i in range(m)]
                self.argmax[:m] = torch.tensor(am, dtype=torch.int32, device=self.device)
                self._rows(m)
                on_chunk(torch.tensor(ids[c0:c0 + m] + [0] * (MAX_DRAFT - m), dtype=torch.int32, device=self.device), self.logits, m, None)
        first = torch.tensor([self.next_token(ids)], dtype=torch.int32, device=self.device)
        session.set_start_token(first)
        self._keep = first
        if not self.with_logits:
            return None
        self.argmax[:1] = first
        self._rows(1)
        return self.logits[0].clone()

    def forward_tokens(self, session, tokens, relpos, mask_rows, n, L):
        """granular verify: the session already holds this draft (DraftModel.lookup installed it)."""
        if not self.with_logits:
            raise SamdError("ScriptedVerifier(with_logits=True) is required for the granular decode()")
        session.set_cache_length(L)
        self.verify(session, MAX_DRAFT)
        return self.logits[:n].clone()

    def verify(self, session, R):
        session.scripted_argmax(self.target, len(self.host_target), self.vocab, self.argmax)
        if self.with_logits:
            self._rows(MAX_DRAFT)
        return dict(argmax=self.argmax, logits=self.logits)

    def compact(self, session):
        pass

    def bucket(self, n):
        return MAX_DRAFT


class ScriptedAcceptance:
    """bench.py's acceptance model: the FULL verify forward of `runner` runs (every kernel, every weight byte), then the
    per-node arg-max is replaced by the next token of a seeded target stream (samd_scripted_argmax).  Without model
    weights or Spec-Bench on the box this is what gives the run realistic, reproducible accept lengths; a random-init
    LM on its own decodes degenerate loops.  The target buffer has a fixed address and length so the step stays one
    hipGraph across requests."""

    def __init__(self, runner, vocab, target_len, ranked_logits=False, order1_hot_vocab=0):
        self.runner, self.vocab, self.target_len = runner, int(vocab), int(target_len)
        # ranked_logits: the verify rows also carry the source's plausible continuations below the scripted arg-max
        # (samd_scripted_logits) -- for plugins that learn from the logits (Token Recycle's top-8 table).  order1_hot_vocab > 0: the
        # source's next token depends on the last token alone (eight successors per token over [3, hot_vocab): bench._succ1)
        self.ranked_logits = bool(ranked_logits)
        self.order1_hot_vocab = int(order1_hot_vocab)
        self.target = torch.zeros(self.target_len, dtype=torch.int32, device=runner.device)
        self.pf_mask = runner.pf_mask
        self._prompt_len = 0

    def set_target(self, target):
        t = torch.as_tensor(np.asarray(target, dtype=np.int32))[:self.target_len]
        self.target.zero_()
        self.target[:t.numel()] = t.to(self.target.device)

    def prefill(self, session, input_ids, on_chunk=None):
        last = self.runner.prefill(session, input_ids, on_chunk)
        n = input_ids.numel()
        session.set_start_token(self.target[n:n + 1])          # the scripted LM's continuation of the prompt
        return last

    def verify(self, session, R):
        b = self.runner.verify(session, R)
        session.scripted_argmax(self.target, self.target_len, self.vocab, b["argmax"])
        if self.ranked_logits:
            if self.order1_hot_vocab > 0:
                session.scripted_logits(b["argmax"], b["logits"], self.order1_hot_vocab, order=1)
            else:
                session.scripted_logits(b["argmax"], b["logits"], self.vocab)
        return b

    def compact(self, session):
        self.runner.compact(session)

    def bucket(self, n):
        return self.runner.bucket(n)

    def warm(self, R):
        self.runner.warm(R)

    def hidden_rows(self, R):
        return self.runner.hidden_rows(R)


class _StopForward(Exception):
    """raised by the engine's layer hook to end a forward after its first launches (the eager head of a step)"""


class _GraphChain:
    """a decode step captured as several hipGraphs replayed back to back on one stream (DecodeEngine._capture), optionally behind an
    EAGER HEAD (SAMD_EAGER_HEAD_LAYERS, off by default: measured slower): the step's first launches (embedding, RoPE rows, the first
    decoder layers) issued directly instead of through hipGraphLaunch, which hands the GPU its first packet only after ~25-30 us."""

    def __init__(self, n, head=None):
        self.parts = [torch.cuda.CUDAGraph() for _ in range(n)]
        self.head = head

    def replay(self):
        if self.head is not None:
            self.head()
        for g in self.parts:
            g.replay()


class DecodeEngine:
    def __init__(self, verifier, session: Session, static: StaticAutomaton, params: Params, recycle: TokenRecycleTable = None,
                 recycle_parent=None, use_graphs=True):
        require_gpu()
        self.verifier, self.session, self.static, self.params = verifier, session, static, params
        self.recycle = recycle
        self.device = torch.device("cuda")
        if recycle is not None:
            self.tr_parent = torch.tensor(recycle_parent, dtype=torch.int32, device=self.device)
            self.tr_tokens = torch.zeros(MAX_DRAFT, dtype=torch.int32, device=self.device)
        self.report_buf = torch.zeros(REPORT_INTS, dtype=torch.int32).pin_memory()
        self._report_np = self.report_buf.numpy()
        self.use_graphs = use_graphs
        self._graphs = {}
        self.bucket_steps = {}               # row bucket -> decode steps run on it (bench.py: bucket_histogram)
        self._views = session.device_views()
        self._n_ptr = self._views["dmeta"] + 4
        # PUSHED REPORT (round 4, include/samd_hip.h): the step kernel writes the report into host-coherent memory itself and the host
        # polls its sequence number -- no D2H copy node, no stream synchronisation behind the cache compaction; the next step's graphs
        # are launched while the compaction still runs.  The plain SAM-only engine only: plugins change the draft after the step kernel
        # (Token Recycle's install) or read the report at their own pace.  SAMD_REPORT_PUSH=0: the copy + synchronise of rounds 1-3.
        self._push = None
        if recycle is None and type(self) is DecodeEngine and os.environ.get("SAMD_REPORT_PUSH", "1") != "0":
            self._push = session.report_target()

    # ---- pieces ---------------------------------------------------------------------------------------
    def _recycle_update(self, d_tokens, logits, n_rows, d_n):
        dt = F32 if logits.dtype == torch.float32 else torch_dtype_code(logits.dtype)
        self.recycle.update(d_tokens, logits, dt, n_rows, logits.shape[1], logits.stride(0), d_n)

    def _install_tree(self):
        """samd/draft.py:63: when the lookup deferred to the tree model, fill and install its draft."""
        self.recycle.draft(self._views["start_token"], self.tr_tokens)
        self.session.set_draft_if_deferred(self.tr_tokens, self.tr_parent, self.recycle.n_nodes, reverse=True)

    def _enqueue_step(self, R):
        b = self.verifier.verify(self.session, R)
        if self.recycle is not None:
            self._recycle_update(self._views["tokens"], b["logits"], min(R, MAX_DRAFT), C.c_void_p(self._n_ptr))
        self.session.step(self.static, self.params, b["argmax"])
        if self.recycle is not None:
            self._install_tree()
        self.verifier.compact(self.session)
        if self._push is None:
            self.session.report_async(self.report_buf)

    def _ingest_beside(self, ids, prefill):
        """DraftModel.update(prompt) -- dyn add_tokens + static transfer_tokens (SO/draft.py:62-67; 2.8 us per token, one wavefront) --
        on a second stream WHILE the LM prefills the prompt: the two touch disjoint state (automata + cursors vs KV cache, cache length,
        start token), and the first lookup waits for both.  SAMD_INGEST_STREAM=0: serially on the decode stream, as round 2 did."""
        s = self.session
        cur = torch.cuda.current_stream()
        if os.environ.get("SAMD_INGEST_STREAM", "1") == "0":
            prefill()
            s.add_tokens(ids)
            s.static_walk(self.static, ids, ids.numel(), commit=True)
            return
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(cur)                           # the session reset and `ids` are ordered before the ingest
        with torch.cuda.stream(self._side):
            s.add_tokens(ids)
            s.static_walk(self.static, ids, ids.numel(), commit=True)
        ids.record_stream(self._side)
        prefill()
        cur.wait_stream(self._side)

    # ---- public ---------------------------------------------------------------------------------------
    def start(self, input_ids):
        """DraftModel.reset + SamdModel.prefill (SO/samd_model.py:96-114) + the first lookup of the decode loop."""
        s = self.session
        s.reset()
        on_chunk = None
        if self.recycle is not None:
            on_chunk = lambda toks, logits, n, hidden: self._recycle_update(toks, logits, n, None)
        ids = input_ids.reshape(-1).to(device=self.device, dtype=torch.int32)
        self._ingest_beside(ids, lambda: self.verifier.prefill(s, ids, on_chunk))
        s.draft(self.static, self.params, self._views["start_token"])
        if self.recycle is not None:
            self._install_tree()
        s.report_async(self.report_buf)
        torch.cuda.current_stream().synchronize()
        return StepReport(self._report_np)

    def step(self, n_next):
        """one decode step on the current draft of n_next nodes; returns the report after it."""
        R = self.verifier.bucket(n_next)
        self.bucket_steps[R] = self.bucket_steps.get(R, 0) + 1
        push = self._push
        g = None
        if self.use_graphs:
            g = self._graphs.get(R)
            if g is None:
                g = self._capture(R)                      # (before the sequence number is read: a capture warms the bucket with real launches)
        if push is not None:
            last = int(push[0][REPORT_INTS])
        if g is None:
            self._enqueue_step(R)
        else:
            g.replay()
        if push is not None:
            from . import lib
            # the pushed report needs no stream synchronisation, so a device fault AFTER the step kernel (cache compaction, the next forward)
            # would surface at some unrelated later call: every 64 steps ask the stream (hipStreamQuery: not-ready is fine, an error raises here)
            self._steps_since_query = getattr(self, "_steps_since_query", 0) + 1
            if self._steps_since_query >= 64:
                self._steps_since_query = 0
                torch.cuda.current_stream().query()
            if lib().samd_report_wait(push[1], last, 5_000_000) != 0:
                torch.cuda.current_stream().synchronize()            # surfaces a device fault; a healthy step has pushed by now
                if int(push[0][REPORT_INTS]) == last:
                    raise SamdError("the step kernel did not push its report")
            self._report_np[:] = push[0][:REPORT_INTS]
            return StepReport(self._report_np)
        torch.cuda.current_stream().synchronize()
        return StepReport(self._report_np)

    # ---- sampling (SO/utils.py:66-104 gen_candidates with greedy=False, :142-184 eval_posterior's sampling branch) ------------------
    # The whole step stays on the device: candidates gathered from the session's draft block, HF's warpers + softmax over the <= 64
    # node rows, the accept / reject walk in samd_posterior_sampled_nodes, the next start token drawn by torch.multinomial, and
    # samd_session_step_given (accept the given verdict, update both automata, next lookup) -- ONE host synchronisation per step (the
    # report + how many uniforms of the host's `random` stream the walk consumed), where the granular decode() makes four.
    def supports_fused_sampling(self):
        return self.recycle is None and type(self) is DecodeEngine

    def start_sampled(self, input_ids):
        s = self.session
        s.reset()
        ids = input_ids.reshape(-1).to(device=self.device, dtype=torch.int32)
        last = []
        self._ingest_beside(ids, lambda: last.append(self.verifier.prefill(s, ids, None)))
        if last[0] is None:
            raise SamdError("this verifier does not expose logits; sampling needs them")
        sample_p = torch.softmax(last[0].reshape(1, -1).float(), dim=-1)         # SamdModel.prefill's return when not greedy
        self._start = self._draw(sample_p)                                        # gen_candidates, utils.py:84
        s.set_start_token(self._start)
        s.draft(self.static, self.params, self._views["start_token"])
        s.report_async(self.report_buf)
        torch.cuda.current_stream().synchronize()
        return StepReport(self._report_np)

    def _draw(self, sample_p):
        """the next step's start token.  The reference draws it at the START of a decode step (gen_candidates), the fused step at the
        END of the previous one (it is samd_session_step_given's lookup key) -- so the draw made by the last step of a request is one
        the reference never makes: the generator state before each draw is kept (host-side: seed + offset, no device round trip)
        and undo_pending_draw() puts it back when the request ends."""
        self._rng_before_draw = torch.cuda.get_rng_state(self.device)
        return torch.multinomial(sample_p, 1).reshape(-1).to(torch.int32)

    def undo_pending_draw(self):
        state = getattr(self, "_rng_before_draw", None)
        if state is not None:
            torch.cuda.set_rng_state(state, self.device)
            self._rng_before_draw = None

    def step_sampled(self, rep, gen_config):
        import random
        s = self.session
        n, C_, D = rep.n, max(rep.n_leaves, 1), max(rep.max_depth, 1)
        R = self.verifier.bucket(n)
        self.bucket_steps[R] = self.bucket_steps.get(R, 0) + 1
        b = self.verifier.verify(s, R)
        node_logits = b["logits"][:n]
        V = node_logits.shape[-1]
        if getattr(self, "_samp", None) is None or self._samp["work"].dtype != node_logits.dtype or self._samp["work"].numel() != V:
            self._samp = dict(cand=torch.zeros(MAX_DRAFT * MAX_DRAFT, dtype=torch.int64, device=self.device),
                              rowmap=torch.zeros(MAX_DRAFT * MAX_DRAFT, dtype=torch.int32, device=self.device),
                              work=torch.empty(V, dtype=node_logits.dtype, device=self.device),
                              out=torch.zeros(5, dtype=torch.int32, device=self.device),
                              out_host=torch.zeros(5, dtype=torch.int32).pin_memory())
        t = self._samp
        s.candidates(t["cand"], t["rowmap"])
        probs = torch.softmax(gen_config.logits_processor(None, node_logits), dim=-1).contiguous()
        n_u = C_ * D + 8
        state = random.getstate()                                                 # the RNG contract of samd_posterior_sampled
        u = torch.from_numpy(np.asarray([random.random() for _ in range(n_u)], dtype=np.float64)).to(self.device)
        from . import _ptr, check, current_stream, lib
        check(lib().samd_posterior_sampled_nodes(_ptr(probs), torch_dtype_code(probs.dtype), _ptr(t["rowmap"]), n, _ptr(t["cand"]), C_, D, V,
                                                 _ptr(u), n_u, _ptr(t["work"]), _ptr(t["out"]), current_stream()))
        out = t["out"]
        cell = out[0].long() * D + out[1].long() - 1
        node = t["rowmap"][cell].long()
        node = torch.where((node < 0) | (node >= n), torch.full_like(node, n - 1), node)
        p_raw = torch.softmax(node_logits[node], dim=0)                           # utils.py:178: the RAW logits of the accepted node
        sample_p = torch.where(out[3] != 0, t["work"], p_raw)                     # utils.py:176-177: the residual after a final rejection
        self._start = self._draw(sample_p.view(1, -1))
        s.step_given(self.static, self.params, out, self._start)
        self.verifier.compact(s)
        s.report_async(self.report_buf)
        t["out_host"].copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        used, status = int(t["out_host"][2]), int(t["out_host"][4])
        random.setstate(state)
        for _ in range(used):
            random.random()
        if status:
            raise SamdError("samd_posterior_sampled ran out of uniforms")
        return StepReport(self._report_np)

    def _capture(self, R):
        """capture the step for row bucket R.  Capture records the launches without executing them, so the request's
        state is untouched; the verifier first runs the bucket once with n = 0 rows (no K/V write, every query row
        masked) so that library handles / workspaces exist before capture."""
        self._warm(R)
        torch.cuda.current_stream().synchronize()
        runner = getattr(self.verifier, "runner", self.verifier)
        cuts = [int(x) for x in os.environ.get("SAMD_GRAPH_SPLIT_LAYER", "2").split(",") if x.strip()]
        n_layers = len(getattr(runner, "w", {}).get("layers", ())) if hasattr(runner, "layer_hook") else 0
        # SAMD_EAGER_HEAD_LAYERS (default 0 = everything in the graphs): embedding + RoPE rows + this many decoder layers launched
        # directly at every step, the graphs hold the rest.  Measured and NOT adopted (round 4, scripts/host_turnaround.py): the GPU gets
        # its first packet ~20 us earlier, but eight Python-issued launches arrive ~8 us apart and the short kernels among them (5 us)
        # leave bubbles -- 3053 vs 3028 us of wall per step with one eager layer.
        eager = int(os.environ.get("SAMD_EAGER_HEAD_LAYERS", "0"))
        eager = eager if 0 < eager < n_layers else 0
        cuts = sorted({c for c in cuts if eager < c < n_layers})
        if not cuts and not eager:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue_step(R)
        else:
            # SEVERAL graphs per step: hipGraphLaunch builds every packet of a graph before the GPU sees the first (~100 us for the step's
            # ~265 nodes, during which the GPU idles); a short head graph (the first layers) starts the GPU early and the rest is
            # enqueued while it runs (scripts/host_turnaround.py).  SAMD_GRAPH_SPLIT_LAYER = the layers to cut before.
            def head():
                def stop(li):
                    if li >= eager:
                        raise _StopForward()
                runner.layer_hook = stop
                try:
                    self.verifier.verify(self.session, R)
                except _StopForward:
                    pass
                finally:
                    runner.layer_hook = None
            g = _GraphChain(len(cuts) + 1, head if eager else None)
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                part = [0]
                started = [not eager]
                if not eager:
                    g.parts[0].capture_begin()

                def hook(li):
                    if not started[0]:
                        if li == eager:                                      # the launches before this layer ran directly (harmless: they
                            g.parts[0].capture_begin()                       # write rows past the committed cache length); capture from here
                            started[0] = True
                        return
                    if li in cuts and part[0] < len(g.parts) - 1:            # a second pass over the layers must not index past the parts
                        g.parts[part[0]].capture_end()
                        part[0] += 1
                        g.parts[part[0]].capture_begin()
                runner.layer_hook = hook
                try:
                    self._enqueue_step(R)
                except BaseException:
                    # never leave the stream in capture mode: end the open part (its graph is discarded with `g`) before re-raising
                    runner.layer_hook = None
                    try:
                        if started[0]:
                            g.parts[part[0]].capture_end()
                    except Exception:
                        pass
                    cur.wait_stream(side)
                    raise
                finally:
                    runner.layer_hook = None
                g.parts[part[0]].capture_end()
            cur.wait_stream(side)
        self._graphs[R] = g
        return g

    def _warm(self, R):
        v = self.verifier
        if hasattr(v, "warm"):
            v.warm(R)


class TreeModelEngine(DecodeEngine):
    """DecodeEngine for tree-draft plugins without device-side hooks (EAGLE-2: samd/tree_model/eagle2.py).  The verify
    forward + accept + SAM update + lookup stay one hipGraph; afterwards the host reads the report, hands the accepted
    tokens and their last hidden states to the plugin (samd/samd_model.py:203-208) and, when the lookup deferred to the
    plugin (samd/draft.py:63), runs its expansion on the device and installs the draft (tokens + parent array; mask,
    positions and retrieve rows come from the tree-buffer kernel)."""

    def __init__(self, verifier, session, static, params, tree_model, use_graphs=True):
        super().__init__(verifier, session, static, params, use_graphs=use_graphs)
        self.tm = tree_model

    def _report(self):
        self.session.report_async(self.report_buf)
        torch.cuda.current_stream().synchronize()
        return StepReport(self._report_np)

    def _maybe_tree(self, rep, start_token=None):
        if rep.type != 2:
            return rep
        if start_token is None:                                  # after the prefill: the deferred draft's root is the start token
            start_token = self.session.read_draft().tokens[0]
        start = torch.tensor([start_token], dtype=torch.long, device=self.device)
        tokens, parents = self.tm.gen_draft_device(start)
        return self._install(rep, tokens, parents)

    def _install(self, rep, tokens, parents):
        """install the plugin's draft (tokens + parent array; the tree-buffer kernel derives mask, positions, retrieve rows).  The
        host knows everything the caller reads from the report afterwards -- the draft is a tree of tokens.numel() nodes -- so it
        does not synchronise again (SAMD_TREE_REPORT_SYNC=1: read the report back, as round 1 did)."""
        self._keep = (tokens.to(torch.int32).contiguous(), parents.to(torch.int32).contiguous())
        self.session.set_draft(self._keep[0], self._keep[1], int(tokens.numel()), type_=1)
        if os.environ.get("SAMD_TREE_REPORT_SYNC", "0") == "1":
            return self._report()
        rep.type, rep.n, rep.n_leaves, rep.max_depth = 1, int(tokens.numel()), -1, -1
        return rep

    def start(self, input_ids):
        s = self.session
        s.reset()
        self.tm.reset()
        ids = input_ids.reshape(-1).to(device=self.device, dtype=torch.int32)
        hidden = []
        self._ingest_beside(ids, lambda: self.verifier.prefill(s, ids, lambda toks, logits, n, h: hidden.append(h[:n].clone())))
        self.tm.update(tokens=ids.to(torch.long), last_hidden_states=torch.cat(hidden, dim=0))
        s.draft(self.static, self.params, self._views["start_token"])
        return self._maybe_tree(self._report())

    def step(self, n_next):
        R = self.verifier.bucket(n_next)
        self.bucket_steps[R] = self.bucket_steps.get(R, 0) + 1
        if not self.use_graphs:
            self._enqueue_step(R)
        else:
            g = self._graphs.get(R) or self._capture(R)
            g.replay()
        torch.cuda.current_stream().synchronize()
        rep = StepReport(self._report_np)
        fast = getattr(self.tm, "gen_draft_from_step", None)
        if rep.type == 2 and fast is not None:
            # the plugin drafts next and nothing is pending: the accepted tokens, their verify rows and the bonus token are read
            # where the step left them on the device; the installed draft's size is known, so the host does not wait for it
            out = fast(self.verifier.hidden_rows(R), self._views, rep.accept, n_next)
            if out is not None:
                return self._install(rep, *out)
        rows = [k if k >= 0 else n_next - 1 for k in rep.kv_index]       # -1 padding selects the last tree node (SO/samd_model.py:144)
        hs = self.verifier.hidden_rows(R)[torch.tensor(rows, dtype=torch.long, device=self.device)]
        self.tm.update(tokens=torch.tensor(rep.tokens, dtype=torch.long, device=self.device), last_hidden_states=hs)
        return self._maybe_tree(rep, rep.next_token)            # the accept step left the bonus token as the next start token
