"""LlamaRunner -- the verify forward of SAM-Decoding on MI355X.

The reference runs HuggingFace `LlamaForCausalLM` with two monkey patches (samd_sam_only/model_patch/llama.py:35-109,
:112-202) and a static KV cache (samd_sam_only/cache.py:37-133).  Here the decoder loop is our own: library GEMMs
(torch.mm -> hipBLASLt) between hand-written gfx950 kernels (embedding gather, RMSNorm+residual, RoPE + KV write at a
device-side offset, tree-mask attention, SiLU*up, row arg-max).  Every dynamic scalar of a decode step -- cache length
L, draft size n, tree depths, tree mask -- is read from device memory (the session's draft block), so one step is a
fixed launch sequence that is captured once per row bucket into a hipGraph.

There is no CPU path: constructing a runner without a GPU raises.
"""
import ctypes as C
import math
import os

import torch

from . import (F16, BF16, MAX_DRAFT, TILE_ROWS, SamdError, Session, Warm, _ptr, check, current_stream, lib, require_gpu,
               torch_dtype_code)


def _cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


class LlamaShape:
    """the architecture numbers the runner needs (subset of transformers.LlamaConfig)."""

    def __init__(self, cfg):
        self.hidden = int(_cfg_get(cfg, "hidden_size"))
        self.inter = int(_cfg_get(cfg, "intermediate_size"))
        self.layers = int(_cfg_get(cfg, "num_hidden_layers"))
        self.heads = int(_cfg_get(cfg, "num_attention_heads"))
        kv = _cfg_get(cfg, "num_key_value_heads")
        self.kv_heads = int(kv) if kv is not None else self.heads
        hd = _cfg_get(cfg, "head_dim")
        self.head_dim = int(hd) if hd else self.hidden // self.heads
        self.vocab = int(_cfg_get(cfg, "vocab_size"))
        self.eps = float(_cfg_get(cfg, "rms_norm_eps", 1e-6))
        self.max_pos = int(_cfg_get(cfg, "max_position_embeddings", 2048))
        rp = _cfg_get(cfg, "rope_parameters") or _cfg_get(cfg, "rope_scaling") or {}
        theta = _cfg_get(cfg, "rope_theta")
        if theta is None and isinstance(rp, dict):
            theta = rp.get("rope_theta")
        self.rope_theta = float(theta if theta is not None else 10000.0)
        self.rope_scaling = dict(rp) if isinstance(rp, dict) else {}
        if self.head_dim != 128:
            raise SamdError("the gfx950 tree-attention kernel is specialised for head_dim 128 (Vicuna-7B / Llama-3-8B)")

    def inv_freq(self):
        """rotary inverse frequencies incl. the 'llama3' scaling rule (what HF's ROPE_INIT_FUNCTIONS computes)."""
        d = self.head_dim
        inv = 1.0 / (self.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        rs = self.rope_scaling
        kind = rs.get("rope_type", rs.get("type", "default")) if rs else "default"
        if kind == "llama3":
            factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
            old = rs["original_max_position_embeddings"]
            wavelen = 2 * math.pi / inv
            scaled = torch.where(wavelen > old / lo, inv / factor, inv)
            smooth = (old / wavelen - lo) / (hi - lo)
            mid = (1 - smooth) * inv / factor + smooth * inv
            is_mid = (wavelen <= old / lo) & (wavelen >= old / hi)
            inv = torch.where(is_mid, mid, scaled)
        elif kind == "linear":
            inv = inv / rs["factor"]
        elif kind not in ("default", None):
            raise SamdError(f"unsupported rope scaling '{kind}'")
        return inv


class LlamaRunner:
    """Own decoder loop over Llama weights resident in HBM.  One instance = one model replica on one GPU."""

    # row buckets of a decode step (one hipGraph each).  The streaming GEMM's cost goes by 16-row tiles (1 / 8 / 16 rows share the
    # 16-row tile), so the buckets above 16 follow its tiles: a 33..48-node draft (match length 8..11 at alpha 4) does not pay for 64 rows
    # 128 (round 5): drafts of 65..128 nodes (max_predicts / n_predicts above 64, which the reference accepts: SO/sam/static_sam.py:183) run as
    # two 64-row tiles of the attention kernel; their projections go to the library GEMM (the streaming kernels' largest row tile is 64, and a
    # 128-row product is no longer a weight stream with a little MFMA attached), so this bucket needs the row-major matrices
    BUCKETS = (1, 8, 16, 32, 48, 64, 128)

    def __init__(self, shape, weights, max_cache_len, dtype=torch.float16, device="cuda", kv=None, native_gemm=True, packed_lm_head=None,
                 attention=None):
        require_gpu()
        self.shape, self.dtype, self.device = shape, dtype, torch.device(device)
        self.dt = torch_dtype_code(dtype)
        if self.dt not in (F16, BF16):
            raise SamdError("LlamaRunner computes in fp16 or bf16")
        s = shape
        self.w = weights
        # samd_gemm_skinny streams the weights itself where the shape allows (N % 128 == 0, K % 256 == 0); a projection that
        # does not fit (say a fine-tune's 32001-row lm_head) goes to the library GEMM on its own, the others keep the kernel
        streams = lambda t: bool(native_gemm) and t.shape[0] % 128 == 0 and t.shape[1] % 256 == 0
        self.native_gemm_max_rows = int(os.environ.get("SAMD_NATIVE_GEMM_MAX_ROWS", 64))     # tuning knob; see forward_rows
        # L2 warm-up (csrc/warm_device.h): the glue launch in front of a projection also reads the first KiB of every workgroup's
        # weight stream into the consuming XCD's L2 while HBM idles.  KiB per projection workgroup; 0 = off, the default: measured
        # zero-sum (profiles/r03_l2_warm.md -- the projections get faster by what the glue launches get slower).
        self.layer_hook = None                                   # callable(layer index), called before a layer's launches (engine: graph split)
        self.warm_kb = int(os.environ.get("SAMD_L2_WARM_KB", 0))
        self.warm_delay = int(os.environ.get("SAMD_L2_WARM_DELAY", 0))        # x 64 cycles before the warm workgroups' first load
        self.warm_where = int(os.environ.get("SAMD_L2_WARM_WHERE", 0))        # output projection: 0 = from the attention splits, 1 = from their merge
        # the attention block of a layer (profiles/r02_attention_variants.md has the per-layer times at Vicuna-7B head geometry):
        #   "split"  = samd_rope_kv_write_cs (per-row cos | sin prepared once per forward: one memory round trip instead of two),
        #              samd_tree_attention(_vt) over 16 KV splits, its merge -- three launches (two with the projection's RoPE epilogue); V cached
        #              transposed since round 6 (see v_layout_t below), row-major before;
        #   "split3" = the round-1 form of the same: RoPE from the position tables (kept for A/B);
        #   "split2" = samd_tree_attention_rope: the splits rotate their own Q rows, one more workgroup owns the n new keys (RoPE, K/V
        #              row write), then the merge of the 17 slots -- two launches; measured SLOWER than three (every split redoes the
        #              rotation; the prologue sits in front of every workgroup's first MFMA): 18.7 vs 16.7 us at 16 rows, 30 vs 23 at 64;
        #   "block"  = samd_attention_block: one launch, V cached transposed, masks with a visible prefix.  One workgroup per head pays
        #              a memory round trip per 512 keys where the splits run side by side (22.8 vs 16.8 us at L = 800), so the base
        #              model's verify does not use it; a draft head's tree levels need the visible prefix and do (one layer).
        self.attention = attention or os.environ.get("SAMD_ATTENTION", "split")
        if self.attention not in ("split", "split2", "split3", "block"):
            raise SamdError(f"unknown attention mode '{self.attention}'")
        # round 6: "split" / "split3" keep V TRANSPOSED too ([H_kv][D][max_len]) wherever the cache length allows 16-byte loads along it: at <= 16 rows
        # samd_tree_attention_vt then runs one wave per (head, KV split) that feeds its MFMAs straight from the V^T rows -- no LDS staging, no
        # workgroup barrier, 11.9 -> 11.0 us per layer at 8 rows (profiles/r06_attention.md).  SAMD_V_LAYOUT=rows keeps the row-major cache (A/B).
        self.v_layout_t = self.attention == "block" or (self.attention in ("split", "split3") and s.head_dim == 128
                                                        and os.environ.get("SAMD_V_LAYOUT", "t") != "rows")
        self.v_transposed = self.v_layout_t
        # second copy of the projection weights in the streaming kernel's packed layout (samd_gemm_pack_weights): the
        # row-major originals stay for the wide prefill's library GEMMs.  2 x 13.5 GB for a 7B model -- HBM capacity
        # (288 GB) is not what this path is short of, bandwidth is.
        def pack(t):
            if not streams(t):
                return None
            out = torch.empty_like(t)
            check(lib().samd_gemm_pack_weights(_ptr(t), _ptr(out), t.shape[0], t.shape[1], current_stream()))
            return out
        def pack_gate_up(t):
            """gate|up rows interleaved in groups of 16 (pair p = gate rows 16p.., up rows 16p..) in the group-major layout of
            samd_gemm_pairs_silu: silu(gate) * up in the projection's epilogue, the pairs dealt out evenly over one workgroup per CU"""
            if s.inter % 16 != 0 or not streams(t):
                return None
            gate, up = t[:s.inter].view(s.inter // 16, 16, -1), t[s.inter:].view(s.inter // 16, 16, -1)
            w = torch.stack([gate, up], dim=1).reshape(2 * s.inter, -1).contiguous()
            out = torch.empty_like(w)
            check(lib().samd_gemm_pack_groups(_ptr(w), _ptr(out), 2 * s.inter, w.shape[1], current_stream()))
            return out
        def pack_qkv64(t):
            """q|k|v in the tile layout of samd_gemm_qkv_rope (RoPE + K/V row write as the projection's epilogue, no split-K; 48- or
            64-column tiles, the library's choice): taken when the launch then has enough workgroups to stream -- >= 128 tiles of 64
            columns' worth (Vicuna-7B: 192 x 64 = 256 x 48 columns; a GQA model like Llama-3-8B has 96 and keeps the split-K
            projection + samd_rope_kv_write_cs)"""
            heads_total = s.heads + 2 * s.kv_heads
            mode = os.environ.get("SAMD_QKV_FUSED", "1")
            # enough workgroups: 128 tiles of 64 columns, or of 48 where 48 divides the matrix (Llama-3-8B: 6144 = 128 x 48 -- alone the
            # fused launch only equals split-K + k_rope_kv there, but it is what the norm-fold forward builds on)
            n_cols = heads_total * 128
            enough = n_cols // 64 >= 128 or (n_cols % 48 == 0 and n_cols // 48 >= 128)
            if (not streams(t) or self.attention != "split" or (not enough and mode != "force") or s.head_dim != 128 or mode == "0"):
                return None
            out = torch.empty_like(t)
            check(lib().samd_gemm_pack_qkv64(_ptr(t), _ptr(out), heads_total, t.shape[1], current_stream()))
            return out
        def pack_groups(t, fold):
            """o_proj / down_proj in the group-major layout of samd_gemm_cs_residual (the norm-fold forward at <= 16 rows)"""
            if not fold or t.shape[0] % 16 != 0 or t.shape[1] % 256 != 0:
                return None
            out = torch.empty_like(t)
            check(lib().samd_gemm_pack_groups(_ptr(t), _ptr(out), t.shape[0], t.shape[1], current_stream()))
            return out
        # packed_lm_head: a draft head shares the base model's lm_head, packed copy included
        layers = []
        for l in weights["layers"]:
            lp = dict(wgu=pack_gate_up(l["wgu"]), wqkv64=pack_qkv64(l["wqkv"]))
            # the 128-column packed q|k|v (split-K projection + samd_rope_kv_write_cs) only where the fused tile form does not exist:
            # with it, no launch of this runner ever reads the other (3.2 GB of a 7B model)
            lp["wqkv"] = pack(l["wqkv"]) if lp["wqkv64"] is None else None
            # the norm-fold forward (include/samd_hip.h: samd_gemm_cs_residual ...) needs the fused q|k|v and gate|up forms
            fold = (lp["wqkv64"] is not None and lp["wgu"] is not None and s.hidden <= 8192 and os.environ.get("SAMD_NORM_FOLD", "1") != "0")
            lp["wo_g"], lp["wdown_g"] = pack_groups(l["wo"], fold), pack_groups(l["wdown"], fold)
            # o_proj / down_proj: the split-K kernel of the 32 / 48 / 64-row buckets can stream the norm-fold forward's group-major copy as well
            # (samd_gemm_skinny_groups, round 6; bit-identical), which makes the 128-column-tile copy redundant: -4 GB of a 7B replica for
            # +1.2 us per layer at 32 / 48 rows, +0.2 at 64 (scripts/gemm_groups_ab.py: 26.2 -> 27.4, 27.7 -> 28.9, 30.1 -> 30.4 us for o + down).
            # Speed is the default; SAMD_GEMM_ONE_COPY=1 -- and release_row_major(), the memory-first mode -- keep only the group-major copy.
            both = os.environ.get("SAMD_GEMM_ONE_COPY", "0") != "1"
            for k in ("wo", "wdown"):
                gm_ok = lp[k + "_g"] is not None and l[k].shape[0] % 128 == 0
                lp[k] = pack(l[k]) if (both or not gm_ok) else None
            layers.append(lp)
        self.wp = dict(lm_head=packed_lm_head if packed_lm_head is not None else pack(weights["lm_head"]), layers=layers)
        self.native_gemm = self.wp["lm_head"] is not None or any(v is not None for l in self.wp["layers"] for v in l.values())
        if not self.native_gemm:
            self.wp = None
        self.fused_mlp = self.wp is not None and all(l["wgu"] is not None for l in self.wp["layers"])
        # norm-fold forward at <= 16 rows: RMSNorm applied by the consuming projection, residual add by the producing one (6 launches per
        # layer instead of 8, no split-K partials): scripts/norm_fold_bench.py, profiles/r03_norm_fold.md
        self.norm_fold = (self.wp is not None and self.attention == "split"
                          and all(l.get("wo_g") is not None and l.get("wdown_g") is not None for l in self.wp["layers"]))
        self.scale = 1.0 / math.sqrt(s.head_dim)
        self.row_major_released = False
        self._length_state(max_cache_len, kv)
        if os.environ.get("SAMD_RELEASE_ROW_MAJOR", "0") == "1":
            self.release_row_major()

    def memory_report(self):
        """bytes of HBM the weights take, by form (documented in DESIGN.md section 2): row-major originals (the wide prefill's library
        GEMMs), and the packed forms the streaming kernels read"""
        def nbytes(t):
            return 0 if t is None or t.device.type == "meta" else t.numel() * t.element_size()
        rep = dict(row_major=sum(nbytes(t) for l in self.w["layers"] for t in l.values()) + nbytes(self.w["lm_head"]) + nbytes(self.w["embed"]))
        if self.wp:
            for k in ("wqkv", "wqkv64", "wo", "wo_g", "wgu", "wdown", "wdown_g"):
                rep["packed_" + k] = sum(nbytes(l.get(k)) for l in self.wp["layers"])
            rep["packed_lm_head"] = nbytes(self.wp["lm_head"])
        rep["total"] = sum(rep.values())
        return rep

    def release_row_major(self):
        """drop the row-major projection matrices (13 GB of a 7B model): they serve only the wide prefill's library GEMMs, so the prompt
        then goes through the streaming kernels in 64-row chunks (SAMD_PREFILL=chunked: ~5 ms per 64 tokens instead of ~9 ms per
        512-token prompt).  Only when every projection has its packed forms and the tensors are the runner's own; shapes stay readable
        (meta tensors).  SAMD_RELEASE_ROW_MAJOR=1 does this at construction."""
        if self.row_major_released or not self.wp:
            return self.row_major_released
        if self.native_gemm_max_rows < TILE_ROWS:
            return False                                   # (a row bucket would fall back to the library GEMM, which reads the row-major matrices;
                                                           #  the 128-row bucket always does: drafts above 64 nodes then raise, see forward_rows)
        if self.wp["lm_head"] is None or any(l.get("wgu") is None or (l.get(k) is None and l.get(k + "_g") is None) for l in self.wp["layers"] for k in ("wo", "wdown")) or \
                any(l.get("wqkv") is None and l.get("wqkv64") is None for l in self.wp["layers"]):
            return False
        for l in self.w["layers"]:
            for k in ("wqkv", "wo", "wgu", "wdown"):
                l[k] = torch.empty(l[k].shape, dtype=l[k].dtype, device="meta")
        for lp in self.wp["layers"]:                             # memory first: o / down keep ONE packed copy (the group-major one serves every bucket)
            for k in ("wo", "wdown"):
                if lp.get(k) is not None and lp.get(k + "_g") is not None:
                    lp[k] = None
        self.row_major_released = True                           # (lm_head stays: draft heads and the granular API read it)
        torch.cuda.empty_cache()
        return True

    def _length_state(self, max_cache_len, kv=None):
        """everything that depends on max_cache_len: KV storage, rotary tables, row buffers, prefill staging.  The weights
        (row-major + packed) do not, so a different max_cache_len between generate() calls re-runs only this."""
        s, dtype = self.shape, self.dtype
        self.max_len = int(max_cache_len)
        self.v_transposed = self.v_layout_t
        if self.v_transposed and (self.max_len < 8 or self.max_len % 8 != 0):
            if self.attention == "block":
                raise SamdError(f"max_cache_len {self.max_len} must be a multiple of 8 (16-byte loads along the transposed V cache)")
            self.v_transposed = False                 # the split launches also read a row-major cache: an odd cache length takes that form
        # KV cache: SamdStaticCache's [1, H_kv, max_cache_len, D] per layer (SO/cache.py:75-84), one allocation
        self.kv = self.kv_ptrs = None
        self.bind_cache(kv if kv is not None else
                        torch.zeros((s.layers, 2, s.kv_heads, self.max_len, s.head_dim), dtype=dtype, device=self.device))
        # rotary tables, fp32 [max_pos][D/2]
        max_pos = max(s.max_pos, self.max_len)
        if getattr(self, "rope_rows", 0) != max_pos:
            ang = torch.outer(torch.arange(max_pos, dtype=torch.float64), s.inv_freq())
            self.cos = ang.cos().float().to(self.device).contiguous()
            self.sin = ang.sin().float().to(self.device).contiguous()
            self.rope_rows = max_pos
        if hasattr(self, "_buf"):
            return                                            # row buffers and prefill staging do not depend on the length
        self._buf = {}
        # prefill staging (chunks of TILE_ROWS rows with a causal chain mask); the mask doubles as the SEQUENCE-draft mask of the granular
        # decode(): row i attends rows 0..i -- low words (nodes 0..63) of all MAX_DRAFT rows, then the high words (nodes 64..127)
        self.pf_tokens = torch.zeros(MAX_DRAFT, dtype=torch.int32, device=self.device)
        self.pf_relpos = torch.arange(MAX_DRAFT, dtype=torch.int32, device=self.device)
        lo = [(1 << (min(i, 63) + 1)) - 1 for i in range(MAX_DRAFT)]
        hi = [0 if i < 64 else (1 << (i - 63)) - 1 for i in range(MAX_DRAFT)]
        self.pf_mask = torch.tensor([r - (1 << 64) if r >= (1 << 63) else r for r in lo + hi], dtype=torch.int64, device=self.device)
        self.pf_n = torch.zeros(1, dtype=torch.int32, device=self.device)

    def resize_cache(self, max_cache_len, storage=None):
        """new max_cache_len (and KV storage) under the same weights; captured hipGraphs of the old buffers are the caller's
        to drop (SamdModel.set_cache rebuilds its engine)."""
        self._length_state(max_cache_len, storage)

    def bind_cache(self, storage):
        """use `storage` [layers, 2, H_kv, max_len, D] (e.g. SamdStaticCache.storage) as the KV cache.  storage[l, 0] holds K rows
        [H_kv][max_len][D]; storage[l, 1] holds V rows likewise, or -- attention mode "block" -- V TRANSPOSED, [H_kv][D][max_len] in
        the same bytes (what samd_attention_block reads as its PV operand without staging; kv_rows() gives the logical view)."""
        s = self.shape
        if tuple(storage.shape) != (s.layers, 2, s.kv_heads, self.max_len, s.head_dim) or storage.dtype != self.dtype:
            raise SamdError(f"KV storage shape/dtype mismatch: {tuple(storage.shape)} {storage.dtype}")
        self.kv = storage
        self.kv_ptrs = torch.tensor([storage[l, j].data_ptr() for j in (0, 1) for l in range(s.layers)],
                                    dtype=torch.int64, device=self.device)

    def kv_rows(self, n):
        """(K, V) of the first n cached positions as [layers, H_kv, n, D] tensors (V un-transposed): the reference's
        key_cache / value_cache contents (SO/cache.py:75-84)"""
        s = self.shape
        k = self.kv[:, 0, :, :n]
        if not self.v_transposed:
            return k, self.kv[:, 1, :, :n]
        v = self.kv[:, 1].reshape(s.layers, s.kv_heads, s.head_dim, self.max_len)[:, :, :, :n].transpose(2, 3)
        return k, v

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_hf(cls, lm, max_cache_len, dtype=None, device="cuda", share_weights=None, **kw):
        """weights of a transformers LlamaForCausalLM (what the reference passes as `lm`).  share_weights (default: env
        SAMD_SHARE_HF_WEIGHTS, off): re-point the HF module's q/k/v and gate/up weights at row slices of the runner's concatenated
        matrices -- saves one row-major copy of the model, but the caller's parameters become views of storage the runner owns
        (matters for save_pretrained / in-place edits), so it is opt-in and logged once."""
        dtype = dtype or next(lm.parameters()).dtype
        shape = LlamaShape(lm.config)
        dev = torch.device(device)

        def get(t):
            return t.detach().to(device=dev, dtype=dtype).contiguous()
        m = lm.model
        layers = []
        # The runner reads q|k|v and gate|up as ONE matrix each.  When the HF module already lives on this device in this dtype, its
        # separate projection weights are re-pointed at row slices of the concatenated matrices (same values, the module keeps working),
        # so the model exists once row-major (+ once packed) instead of the HF copy + the concatenated copy + the packed copy.
        # Opt-in (share_weights=True / SAMD_SHARE_HF_WEIGHTS=1): by default the caller's module is left untouched.
        share = (os.environ.get("SAMD_SHARE_HF_WEIGHTS", "0") == "1") if share_weights is None else bool(share_weights)
        shared = [0]

        def fuse(linears):
            ws = [l.weight for l in linears]
            cat = get(torch.cat(ws, dim=0))
            if share and all(w.device == cat.device and w.dtype == cat.dtype for w in ws):
                r = 0
                for l in linears:
                    n = l.weight.shape[0]
                    l.weight.data = cat[r:r + n]
                    r += n
                shared[0] += 1
            return cat
        for lyr in m.layers:
            a, f = lyr.self_attn, lyr.mlp
            for lin in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, f.gate_proj, f.up_proj, f.down_proj):
                if getattr(lin, "bias", None) is not None:
                    raise SamdError("LlamaRunner: projection biases are not supported")
            layers.append(dict(
                wqkv=fuse((a.q_proj, a.k_proj, a.v_proj)),
                wo=get(a.o_proj.weight),
                wgu=fuse((f.gate_proj, f.up_proj)),
                wdown=get(f.down_proj.weight),
                ln1=get(lyr.input_layernorm.weight), ln2=get(lyr.post_attention_layernorm.weight)))
        weights = dict(embed=get(m.embed_tokens.weight), layers=layers, norm=get(m.norm.weight), lm_head=get(lm.lm_head.weight))
        if shared[0]:
            import logging
            logging.getLogger("samd_hip").info("LlamaRunner.from_hf: %d fused projection groups now back the HF module's q/k/v and gate/up "
                                               "weights (share_weights); its parameters are views of the runner's matrices", shared[0])
        return cls(shape, weights, max_cache_len, dtype, device, **kw)

    @classmethod
    def random_init(cls, cfg, max_cache_len, dtype=torch.float16, device="cuda", seed=0, std=0.02, **kw):
        """random-init weights of the given architecture, created directly in HBM (no checkpoint on the box)."""
        require_gpu()
        shape = LlamaShape(cfg)
        g = torch.Generator(device=device).manual_seed(seed)

        def rnd(*size):
            return (torch.randn(size, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
        s = shape
        qkv_out = (s.heads + 2 * s.kv_heads) * s.head_dim
        layers = [dict(wqkv=rnd(qkv_out, s.hidden), wo=rnd(s.hidden, s.heads * s.head_dim), wgu=rnd(2 * s.inter, s.hidden),
                       wdown=rnd(s.hidden, s.inter), ln1=torch.ones(s.hidden, dtype=dtype, device=device),
                       ln2=torch.ones(s.hidden, dtype=dtype, device=device)) for _ in range(s.layers)]
        weights = dict(embed=rnd(s.vocab, s.hidden), layers=layers, norm=torch.ones(s.hidden, dtype=dtype, device=device),
                       lm_head=rnd(s.vocab, s.hidden))
        return cls(shape, weights, max_cache_len, dtype, device, **kw)

    def weight_bytes(self):
        """bytes of weights one decode step streams from HBM (the embedding table is only gathered)."""
        n = self.w["lm_head"].numel() + self.w["norm"].numel()
        for l in self.w["layers"]:
            n += sum(t.numel() for t in l.values())
        return n * self.w["lm_head"].element_size()

    # ------------------------------------------------------------------------------------------------
    def _buffers(self, R):
        if R not in self._buf:
            s, dt, dev = self.shape, self.dtype, self.device
            RP = max(R, 16) if self.native_gemm else R           # the skinny GEMM reads 16 / 32 / 64 rows (pad rows are zero)
            z = lambda r, *sz: torch.zeros((max(r, RP),) + sz, dtype=dt, device=dev)
            ws_bytes = max(lib().samd_tree_attention_workspace(R, s.heads, s.head_dim), lib().samd_tree_attention_rope_workspace(R, s.heads, s.head_dim))
            part_elems = 0
            if self.native_gemm:
                qkv_out = (s.heads + 2 * s.kv_heads) * s.head_dim
                for n, k in ((qkv_out, s.hidden), (s.hidden, s.heads * s.head_dim), (2 * s.inter, s.hidden), (s.hidden, s.inter)):
                    if RP <= self.native_gemm_max_rows:          # (above it every projection is a library GEMM: no split-K partials)
                        part_elems = max(part_elems, lib().samd_gemm_splits(n, k, RP) * RP * n)
            self._buf[R] = dict(x=z(R, s.hidden), h=z(R, s.hidden), qkv=z(R, (s.heads + 2 * s.kv_heads) * s.head_dim),
                                q=z(R, s.heads, s.head_dim), attn=z(R, s.heads, s.head_dim), o=z(R, s.hidden),
                                gu=z(R, 2 * s.inter), act=z(R, s.inter), d=z(R, s.hidden), logits=z(R, s.vocab),
                                argmax=torch.zeros(MAX_DRAFT, dtype=torch.int32, device=dev), rows_pad=RP,
                                part=torch.zeros(max(part_elems, 1), dtype=torch.float32, device=dev),
                                ws=torch.zeros(ws_bytes, dtype=torch.uint8, device=dev), ws_bytes=ws_bytes,
                                cs=torch.zeros((MAX_DRAFT, s.head_dim), dtype=torch.float32, device=dev),
                                ssq=torch.zeros((max(s.hidden // 16, 1), 16), dtype=torch.float32, device=dev))
        return self._buf[R]

    def max_draft_rows(self):
        """the largest draft (nodes) this runner can verify: 128 needs the library GEMM on the row-major matrices and the split tree
        attention's two-tile form; otherwise the streaming kernels' 64-row tile is the limit.  DraftModel parameters are clamped to this
        when a session's engine is made (samd_sam_only.sam._common.clamp_to_verifier), so a wide draft never fails inside forward_rows."""
        wide_ok = (not self.row_major_released) and self.attention in ("split", "split3") and not getattr(self, "draft_head", False)
        return MAX_DRAFT if wide_ok else min(MAX_DRAFT, 64)

    def bucket(self, n):
        for b in self.BUCKETS:
            if n <= b:
                return b
        raise SamdError(f"draft of {n} nodes exceeds {MAX_DRAFT}")

    def forward_rows(self, R, d_tokens, d_relpos, d_mask, d_L, d_n, x_in=None, d_vis=None):
        """one forward over R rows; all of d_* are device pointers (ints / tensors).  Returns the buffers of bucket R
        (logits [R, V], argmax int32[64] with rows < n valid).  x_in [R, hidden]: the rows' input states instead of the
        token embedding (EAGLE draft heads feed fc([embed ; hidden])).  With `self.draft_head` the decoder is an EAGLE head:
        layer 0 has no input norm and lm_head reads the residual stream itself (b["x"] = the head's output states)."""
        L, s, b, dt, st = lib(), self.shape, self._buffers(R), self.dt, current_stream()
        RP, part = b["rows_pad"], b["part"]
        if RP > self.native_gemm_max_rows and self.row_major_released:
            raise SamdError(f"a {R}-row forward needs the row-major projection matrices (released: SAMD_RELEASE_ROW_MAJOR / release_row_major()); "
                            f"drafts above {self.native_gemm_max_rows} nodes and the library-GEMM path are unavailable on this runner")
        if (self.norm_fold and RP == 16 and x_in is None and d_vis is None and not getattr(self, "draft_head", False)
                and RP <= self.native_gemm_max_rows):
            return self._forward_rows_fold(R, b, d_tokens, d_relpos, d_mask, d_L, d_n)

        def hint(w, wp, fused=False, is_head=False):
            """samd_warm_t of the projection (w row-major, wp packed) that follows a glue launch, or None"""
            if wp is None or RP > self.native_gemm_max_rows or self.warm_kb <= 0:
                return None
            n, k = w.shape
            sp = 1 if (fused or is_head) else L.samd_gemm_splits(n, k, RP)
            return C.byref(Warm(wp.data_ptr(), n, k, sp, self.warm_kb, self.warm_delay, self.warm_where))

        def gemm(a, w, wp, out, wg=None):
            """out = a @ w.T (wp = w in the packed 128-column-tile layout, wg = w group-major: whichever exists); returns
            (operand for the consumer, n_partials, partial_stride)."""
            n, k = w.shape
            # measured on MI355X (scripts/forward_ablation.py, whole forward incl. the consumers' partial-sum reads), ours vs
            # the library GEMM: 3.46 vs 4.75 ms at <= 16 rows, 3.74 vs 4.62 at 32, 4.72 vs 5.15 at 64
            if (wp is None and wg is None) or RP > self.native_gemm_max_rows:
                torch.mm(a[:R], w.t(), out=out[:R])
                return out, 0, 0
            sp = L.samd_gemm_splits(n, k, RP) if out is not b["logits"] else 1
            if wp is not None and os.environ.get("SAMD_GEMM_PREFER_GROUPS", "0") != "1":
                check(L.samd_gemm_skinny(_ptr(a), _ptr(wp), RP, n, k, sp, _ptr(part), _ptr(out), dt, st))
            else:
                check(L.samd_gemm_skinny_groups(_ptr(a), _ptr(wg if wg is not None else wp), RP, n, k, sp, _ptr(part), _ptr(out), dt, st))
            return (out, 0, 0) if sp == 1 else (part, sp, RP * n)

        if x_in is None:
            check(L.samd_embed_rows(_ptr(d_tokens), _ptr(self.w["embed"]), _ptr(b["x"]), R, s.hidden, s.vocab, dt, st))
        else:
            rows_in = min(R, x_in.shape[0])
            if x_in.data_ptr() != b["x"].data_ptr():              # a caller may stage the rows in the bucket's own buffer
                b["x"][:rows_in].copy_(x_in[:rows_in])            # rows past d_n are never consumed
        head = getattr(self, "draft_head", False)
        block = self.attention == "block"
        vt = self.v_transposed and not block          # the split launches over a transposed V cache (round 6)
        if self.attention != "split3":
            # cos / sin of every row's position (visible length + relative position), once per forward: the attention launches of all
            # layers read them without first having to wait for L
            check(L.samd_rope_rows(_ptr(d_relpos), _ptr(d_vis if d_vis is not None else d_L), _ptr(self.cos), _ptr(self.sin), _ptr(b["cs"]), R,
                                   s.head_dim, self.rope_rows, st))
        if d_vis is not None and not block:
            raise SamdError("a visible length different from the write position needs attention mode 'block'")
        delta, dn, dstride = None, 0, 0
        packed = self.wp["layers"] if self.wp else [{}] * len(self.w["layers"])
        for li, w in enumerate(self.w["layers"]):
            if self.layer_hook is not None:
                self.layer_hook(li)
            wp = packed[li]
            raw_in = head and li == 0                             # eagle2_model.py:516-519: no input layer-norm in the head's layer
            if not raw_in:
                check(L.samd_rmsnorm_warm(_ptr(b["x"]), _ptr(delta), _ptr(w["ln1"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, dn, dstride,
                                          hint(w["wqkv"], wp.get("wqkv")), st))
            fused_qkv = wp.get("wqkv64") is not None and self.attention == "split" and RP <= self.native_gemm_max_rows and d_vis is None
            if fused_qkv:
                # q|k|v projection + RoPE + K/V row write in one launch (csrc/gemm_kernels.hip: k_gemm_qkv_rope)
                check((L.samd_gemm_qkv_rope_vt if vt else L.samd_gemm_qkv_rope)(
                    _ptr(b["x"] if raw_in else b["h"]), _ptr(wp["wqkv64"]), RP, s.hidden, _ptr(b["cs"]), _ptr(d_L), _ptr(d_n),
                    _ptr(b["q"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), s.heads, s.kv_heads, s.head_dim, self.max_len, dt, st))
            else:
                src, n_p, stride = gemm(b["x"] if raw_in else b["h"], w["wqkv"], wp.get("wqkv"), b["qkv"])
            if block:
                # RoPE + K row / V^T column write + tree attention + merge of the tile partials: one launch (csrc/attn_kernels.hip)
                check(L.samd_attention_block(_ptr(src), n_p, stride, _ptr(b["cs"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), _ptr(b["attn"]), dt, R,
                                             s.heads, s.kv_heads, s.head_dim, self.max_len, _ptr(d_mask), _ptr(d_L), _ptr(d_vis), _ptr(d_n), self.scale, st))
            elif self.attention == "split2":
                check(L.samd_tree_attention_rope(_ptr(src), n_p, stride, _ptr(b["cs"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), _ptr(b["attn"]), dt, R,
                                                 s.heads, s.kv_heads, s.head_dim, self.max_len, _ptr(d_mask), _ptr(d_L), _ptr(d_n), self.scale,
                                                 _ptr(b["ws"]), b["ws_bytes"], st))
            else:
                if fused_qkv:
                    pass
                elif self.attention == "split":
                    check((L.samd_rope_kv_write_cs_vt if vt else L.samd_rope_kv_write_cs)(
                        _ptr(src), _ptr(d_relpos), _ptr(d_L), _ptr(d_n), _ptr(b["cs"]), _ptr(b["q"]), _ptr(self.kv[li, 0]),
                        _ptr(self.kv[li, 1]), R, s.heads, s.kv_heads, s.head_dim, self.max_len, dt, n_p, stride, st))
                else:
                    check((L.samd_rope_kv_write_vt if vt else L.samd_rope_kv_write)(
                        _ptr(src), _ptr(d_relpos), _ptr(d_L), _ptr(d_n), _ptr(self.cos), _ptr(self.sin),
                        _ptr(b["q"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), R, s.heads, s.kv_heads,
                        s.head_dim, self.max_len, self.rope_rows, dt, n_p, stride, st))
                check((L.samd_tree_attention_vt if vt else L.samd_tree_attention_warm)(
                    _ptr(b["q"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), _ptr(b["attn"]), dt, R, s.heads,
                    s.kv_heads, s.head_dim, self.max_len, _ptr(d_mask), _ptr(d_L), _ptr(d_n), self.scale,
                    _ptr(b["ws"]), b["ws_bytes"], hint(w["wo"], wp.get("wo")), st))
            src, n_p, stride = gemm(b["attn"].view(b["attn"].shape[0], -1), w["wo"], wp.get("wo"), b["o"], wg=wp.get("wo_g"))
            check(L.samd_rmsnorm_warm(_ptr(b["x"]), _ptr(src), _ptr(w["ln2"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, n_p, stride,
                                      None, st))           # (no warm-up hint: gate|up is packed group-major, the hint describes 128-column tiles)
            if self.fused_mlp and RP <= self.native_gemm_max_rows:
                check(L.samd_gemm_pairs_silu(_ptr(b["h"]), _ptr(wp["wgu"]), RP, s.inter, s.hidden, _ptr(b["act"]), dt, st))
            else:
                src, n_p, stride = gemm(b["h"], w["wgu"], None, b["gu"])           # wgu is only ever packed for the fused form
                check(L.samd_silu_mul(_ptr(src), _ptr(b["act"]), R, s.inter, dt, n_p, stride, st))
            delta, dn, dstride = gemm(b["act"], w["wdown"], wp.get("wdown"), b["d"], wg=wp.get("wdown_g"))
        check(L.samd_rmsnorm_warm(_ptr(b["x"]), _ptr(delta), _ptr(self.w["norm"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, dn, dstride,
                                  hint(self.w["lm_head"], self.wp["lm_head"] if self.wp else None, is_head=True), st))
        # (for a draft head the call above only folds the last projection into the residual stream; its norm output is unused)
        gemm(b["x"] if head else b["h"], self.w["lm_head"], self.wp["lm_head"] if self.wp else None, b["logits"])
        if not head:                                              # a draft head's callers rank the logits themselves
            check(L.samd_argmax_rows(_ptr(b["logits"]), dt, R, s.vocab, s.vocab, None, _ptr(b["argmax"]), st))
        return b

    def _forward_rows_fold(self, R, b, d_tokens, d_relpos, d_mask, d_L, d_n):
        """forward_rows at <= 16 rows in the norm-fold form: the residual stream b["x"] is complete after every projection that adds to it
        (samd_gemm_cs_residual: complete sums + residual + the rows' sums of squares), and input_layernorm / post_attention_layernorm are
        applied by the q|k|v and gate|up projections on their way into LDS.  Six launches per decoder layer instead of eight."""
        L, s, dt, st = lib(), self.shape, self.dt, current_stream()
        x, ssq = b["x"], b["ssq"]
        rows_cs = 8 if R <= 8 else 16            # the complete-sum projections fetch only the rows a <= 8-node draft has
        rows_a = rows_cs if os.environ.get("SAMD_NORM_ROWS8", "1") != "0" else 16      # ... and so do the norm-applying ones (round 4)
        if s.head_dim == 128 and os.environ.get("SAMD_FUSE_EMBED_ROPE", "1") != "0":
            # embedding rows + their sums of squares and the rows' cos | sin: one launch (neither depends on the other)
            check(L.samd_embed_rows_ssq_rope(_ptr(d_tokens), _ptr(self.w["embed"]), _ptr(x), _ptr(ssq), 16, s.hidden, s.vocab, dt, _ptr(d_relpos), _ptr(d_L),
                                             _ptr(self.cos), _ptr(self.sin), _ptr(b["cs"]), R, s.head_dim, self.rope_rows, st))
        else:
            check(L.samd_embed_rows_ssq(_ptr(d_tokens), _ptr(self.w["embed"]), _ptr(x), _ptr(ssq), 16, s.hidden, s.vocab, dt, st))
            check(L.samd_rope_rows(_ptr(d_relpos), _ptr(d_L), _ptr(self.cos), _ptr(self.sin), _ptr(b["cs"]), R, s.head_dim, self.rope_rows, st))
        attn2d = b["attn"].view(b["attn"].shape[0], -1)
        # round 6: over a transposed V cache the projection's epilogue writes V^T columns and the attention is the one-wave-per-split launch
        qkv_launch = L.samd_gemm_qkv_rope_norm_vt if self.v_transposed else L.samd_gemm_qkv_rope_norm
        attn_launch = L.samd_tree_attention_vt if self.v_transposed else L.samd_tree_attention_warm
        for li, (w, wp) in enumerate(zip(self.w["layers"], self.wp["layers"])):
            if self.layer_hook is not None:
                self.layer_hook(li)
            check(qkv_launch(_ptr(x), _ptr(ssq), _ptr(w["ln1"]), s.eps, _ptr(wp["wqkv64"]), rows_a, s.hidden, _ptr(b["cs"]), _ptr(d_L), _ptr(d_n),
                             _ptr(b["q"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), s.heads, s.kv_heads, s.head_dim, self.max_len, dt, st))
            check(attn_launch(_ptr(b["q"]), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), _ptr(b["attn"]), dt, R, s.heads,
                              s.kv_heads, s.head_dim, self.max_len, _ptr(d_mask), _ptr(d_L), _ptr(d_n), self.scale,
                              _ptr(b["ws"]), b["ws_bytes"], None, st))
            check(L.samd_gemm_cs_residual(_ptr(attn2d), _ptr(wp["wo_g"]), rows_cs, s.hidden, attn2d.shape[1], _ptr(x), _ptr(ssq), dt, st))
            check(L.samd_gemm_pairs_silu_norm(_ptr(x), _ptr(ssq), _ptr(w["ln2"]), s.eps, _ptr(wp["wgu"]), rows_a, s.inter, s.hidden, _ptr(b["act"]), dt, st))
            check(L.samd_gemm_cs_residual(_ptr(b["act"]), _ptr(wp["wdown_g"]), rows_cs, s.hidden, s.inter, _ptr(x), _ptr(ssq), dt, st))
        check(L.samd_rmsnorm(_ptr(x), None, _ptr(self.w["norm"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, 0, 0, st))
        wl = self.wp["lm_head"]
        if wl is not None:
            check(L.samd_gemm_skinny(_ptr(b["h"]), _ptr(wl), 16, s.vocab, s.hidden, 1, _ptr(b["part"]), _ptr(b["logits"]), dt, st))
        else:
            torch.mm(b["h"][:R], self.w["lm_head"].t(), out=b["logits"][:R])
        check(L.samd_argmax_rows(_ptr(b["logits"]), dt, R, s.vocab, s.vocab, None, _ptr(b["argmax"]), st))
        return b

    # ------------------------------------------------------------------------------------------------
    def prefill(self, session: Session, input_ids, on_chunk=None):
        """SamdModel.prefill's LM part (SO/samd_model.py:96-114): the prompt goes through the same kernels in chunks of
        64 rows with a causal chain mask; K/V land at [0, N).  Leaves cache_length = N in the session and the arg-max of
        the last prompt position in session.start_token.  on_chunk(tokens int32[64], logits [64,V], n, hidden [64,H]) is
        called per chunk (Token Recycle learns from the prompt logits, EAGLE-2 from the last hidden states:
        S/samd_model.py:117-122)."""
        ids = input_ids.reshape(-1).to(device=self.device, dtype=torch.int32)
        N = ids.numel()
        if N < 1 or N > self.max_len:
            raise SamdError(f"prompt of {N} tokens does not fit max_cache_len {self.max_len}")
        if N >= 2 * TILE_ROWS and os.environ.get("SAMD_PREFILL", "wide") != "chunked" and not self.row_major_released:
            return self._prefill_wide(session, ids, on_chunk)
        v = session.device_views()
        b = None
        for c0 in range(0, N, TILE_ROWS):
            n = min(TILE_ROWS, N - c0)
            self.pf_tokens.zero_()
            self.pf_tokens[:n] = ids[c0:c0 + n]
            self.pf_n.fill_(n)
            session.set_cache_length(c0)
            b = self.forward_rows(TILE_ROWS, self.pf_tokens, self.pf_relpos, self.pf_mask, v["cache_length"], self.pf_n)
            if on_chunk is not None:
                on_chunk(self.pf_tokens, b["logits"], n, b["h"])
        session.set_cache_length(N)
        session.set_start_token(b["argmax"][(N - 1) % TILE_ROWS:])
        return b["logits"][(N - 1) % TILE_ROWS]

    # ---- the wide prefill's library calls, shaped for what the library does well on 256 CUs (profiles/r05_prefill.md) ----
    PF_SPLIT_MIN_ROWS = 1024          # below this no projection of the prompt is worth splitting (and short prompts keep one code path)
    PF_ATTN_PAD = 128                 # fused causal SDPA runs 40-45 % slower on row counts that are not a multiple of this (r05_gemm_rows.log)

    def _is_gfx950(self):
        if not hasattr(self, "_gfx950"):
            name = getattr(torch.cuda.get_device_properties(self.device), "gcnArchName", "") or ""
            self._gfx950 = name.split(":")[0] == "gfx950"
        return self._gfx950

    def _time_mm(self, x, wt, out, reps=3):
        best = float("inf")
        torch.mm(x, wt, out=out)
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.mm(x, wt, out=out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    def tune_prefill(self, max_rows=None):
        """Where does a projection of the prompt fall off a tile-quantisation cliff?  hipBLASLt's time over the prompt's rows is a staircase
        (q|k|v and gate|up of a 7B layer: 103 / 196 us at 1280 rows, 153 / 298 at 1281-1536 -- a third round of tiles on 256 CUs), while the
        remainder rows alone cost 34-62 us.  For every projection this measures the staircase once -- rows R = 256 k, R + 64, and the small
        products 64..256 -- and keeps, per R, whether `mm(rows[:R]) + mm(rows[R:])` beats one call.  ~0.2 s at 7B shapes, once per runner (lazily on the
        first prompt of >= PF_SPLIT_MIN_ROWS rows, or call it during warm-up); SAMD_PREFILL_SPLIT=0 disables the splits."""
        self._pf_plan = {}
        if os.environ.get("SAMD_PREFILL_SPLIT", "1") == "0" or self.row_major_released:
            return self._pf_plan
        max_rows = min(int(max_rows or self.max_len), 4096)          # longer prompts: one call per projection
        w0 = self.w["layers"][0]
        try:
            for key in ("wqkv", "wo", "wgu", "wdown"):
                wt = w0[key].t()
                K, N = wt.shape
                x = torch.zeros((max_rows + 64, K), dtype=self.dtype, device=self.device)
                out = torch.empty((max_rows + 64, N), dtype=self.dtype, device=self.device)
                small = {r: self._time_mm(x[:r], wt, out[:r]) for r in (64, 128, 192, 256)}
                plan = {}
                for R in range(max(256, (self.PF_SPLIT_MIN_ROWS // 256) * 256), max_rows, 256):
                    t_at = self._time_mm(x[:R], wt, out[:R])
                    t_past = self._time_mm(x[:R + 64], wt, out[:R + 64])
                    # rows in (R, R + 256]: one call costs ~t_past whatever the count (the staircase is flat between steps); two calls t_at + small
                    plan[R] = {r: t_at + small[r] < 0.95 * t_past for r in small}
                self._pf_plan[key] = plan
                del x, out
        except RuntimeError as e:                                    # e.g. no memory for the scratch operands: the plan is an optimisation, not a need
            import warnings
            warnings.warn(f"tune_prefill: measurement failed ({str(e)[:120]}); projections of long prompts stay single library calls", RuntimeWarning)
            self._pf_plan = {}
        return self._pf_plan

    def prefill_plan_summary(self):
        """{projection: [first-call row counts R at which a prompt of R + 1 .. R + 256 rows is split]} -- for logs"""
        plan = getattr(self, "_pf_plan", None) or {}
        return {k: [R for R, d in sorted(v.items()) if any(d.values())] for k, v in plan.items()}

    def _pf_split(self, key, M):
        """row count of the first of two library calls for projection `key` over M prompt rows, or 0 for one call"""
        if M <= self.PF_SPLIT_MIN_ROWS:
            return 0
        if getattr(self, "_pf_plan", None) is None:
            self.tune_prefill()
        R = ((M - 1) // 256) * 256
        rest = -(-(M - R) // 64) * 64
        return R if self._pf_plan.get(key, {}).get(R, {}).get(rest, False) else 0

    def _pf_mm(self, x, w, out, key):
        R = self._pf_split(key, x.shape[0])
        if R:
            torch.mm(x[:R], w.t(), out=out[:R])
            torch.mm(x[R:], w.t(), out=out[R:])
        else:
            torch.mm(x, w.t(), out=out)

    def _prefill_wide(self, session: Session, ids, on_chunk=None):
        """the whole prompt in one pass: compute-bound, so the GEMMs go to the library (N x K x N_out at full MFMA rate) and
        the causal attention to samd_prefill_attention / samd_prefill_attention_vt (round 5; PyTorch's fused SDPA before); norm /
        RoPE + K/V write / SiLU*up / arg-max are our kernels as well.  A
        per-chunk consumer (Token Recycle: the prompt's logits, EAGLE: its last hidden states) gets them afterwards in
        64-row slices of one [N, V] lm_head product.  Round 5: a projection whose row count sits just past a tile-quantisation step of
        the library is issued as two calls (tune_prefill), and the attention runs on the row count padded to a multiple of 128 -- zero
        query rows and zero K / V rows BEHIND the prompt, which causality keeps out of every real row (their own outputs are dropped)."""
        L, s, dt, st, N = lib(), self.shape, self.dt, current_stream(), ids.numel()
        dev, ty = self.device, self.dtype
        z = lambda *sz: torch.empty(sz, dtype=ty, device=dev)
        x, h = z(N, s.hidden), z(N, s.hidden)
        # the prompt's causal attention: our kernel (samd_prefill_attention: any row count, nothing behind the prompt is read) on the row-major
        # or the transposed cache; PyTorch's fused SDPA otherwise (head_dim != 128, SAMD_PREFILL_ATTENTION=sdpa, not a gfx950)
        own_attn = (s.head_dim == 128 and os.environ.get("SAMD_PREFILL_ATTENTION", "own") != "sdpa"
                    and self._is_gfx950())             # the kernel needs 136 KiB of LDS and gfx950's permlane swaps: any other device takes SDPA
        Np = N if own_attn else -(-N // self.PF_ATTN_PAD) * self.PF_ATTN_PAD
        if Np > self.max_len:
            Np = N
        ao = z(N, s.heads * s.head_dim) if own_attn else None
        qkv_p, qp = z(Np, (s.heads + 2 * s.kv_heads) * s.head_dim), z(Np, s.heads, s.head_dim)
        qkv, q = qkv_p[:N], qp[:N]
        if Np > N:
            qp[N:].zero_()
            if self.v_transposed:
                qkv_p[N:].zero_()                     # V comes straight from the projection in this mode: its rows behind the prompt
                self.kv[:, 0, :, N:Np].zero_()
            else:
                self.kv[:, :, :, N:Np].zero_()        # rows behind the prompt: free space of the cache (the first decode steps overwrite them)
        o, gu, act, d = z(N, s.hidden), z(N, 2 * s.inter), z(N, s.inter), z(N, s.hidden)
        relpos = torch.arange(N, dtype=torch.int32, device=dev)
        d_L = torch.zeros(1, dtype=torch.int32, device=dev)
        d_n = torch.full((1,), N, dtype=torch.int32, device=dev)
        check(L.samd_embed_rows(_ptr(ids), _ptr(self.w["embed"]), _ptr(x), N, s.hidden, s.vocab, dt, st))
        delta = None
        for li, w in enumerate(self.w["layers"]):
            check(L.samd_rmsnorm(_ptr(x), _ptr(delta), _ptr(w["ln1"]), _ptr(h), N, s.hidden, s.eps, dt, 0, 0, st))
            self._pf_mm(h, w["wqkv"], qkv, "wqkv")
            if self.v_transposed:
                check(L.samd_rope_kv_write_vt(_ptr(qkv), _ptr(relpos), _ptr(d_L), _ptr(d_n), _ptr(self.cos), _ptr(self.sin), _ptr(q),
                                              _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), N, s.heads, s.kv_heads, s.head_dim, self.max_len,
                                              self.rope_rows, dt, 0, 0, st))                                         # q, K rows, V^T columns of the cache
                vv = qkv_p[:, (s.heads + s.kv_heads) * s.head_dim:].view(Np, s.kv_heads, s.head_dim).transpose(0, 1)  # (SDPA below: V straight from the projection)
            else:
                check(L.samd_rope_kv_write(_ptr(qkv), _ptr(relpos), _ptr(d_L), _ptr(d_n), _ptr(self.cos), _ptr(self.sin), _ptr(q),
                                           _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), N, s.heads, s.kv_heads, s.head_dim, self.max_len,
                                           self.rope_rows, dt, 0, 0, st))
                vv = self.kv[li, 1][:, :Np]
            if own_attn:
                check((L.samd_prefill_attention_vt if self.v_transposed else L.samd_prefill_attention)(
                    _ptr(q), _ptr(self.kv[li, 0]), _ptr(self.kv[li, 1]), _ptr(ao), dt, N, 0, s.heads, s.kv_heads,
                    s.head_dim, self.max_len, self.scale, st))
                self._pf_mm(ao, w["wo"], o, "wo")
            else:
                kk = self.kv[li, 0][:, :Np]
                if s.kv_heads != s.heads:
                    kk, vv = kk.repeat_interleave(s.heads // s.kv_heads, dim=0), vv.repeat_interleave(s.heads // s.kv_heads, dim=0)
                att = torch.nn.functional.scaled_dot_product_attention(qp.transpose(0, 1)[None], kk[None], vv[None], is_causal=True, scale=self.scale)
                self._pf_mm(att[0, :, :N].transpose(0, 1).reshape(N, -1), w["wo"], o, "wo")
            check(L.samd_rmsnorm(_ptr(x), _ptr(o), _ptr(w["ln2"]), _ptr(h), N, s.hidden, s.eps, dt, 0, 0, st))
            self._pf_mm(h, w["wgu"], gu, "wgu")
            check(L.samd_silu_mul(_ptr(gu), _ptr(act), N, s.inter, dt, 0, 0, st))
            self._pf_mm(act, w["wdown"], d, "wdown")
            delta = d
        check(L.samd_rmsnorm(_ptr(x), _ptr(delta), _ptr(self.w["norm"]), _ptr(h), N, s.hidden, s.eps, dt, 0, 0, st))
        b = self._buffers(1)
        if on_chunk is None:
            torch.mm(h[N - 1:N], self.w["lm_head"].t(), out=b["logits"][:1])
        else:
            logits = torch.mm(h, self.w["lm_head"].t())
            for c0 in range(0, N, TILE_ROWS):
                n = min(TILE_ROWS, N - c0)
                on_chunk(ids[c0:c0 + n], logits[c0:c0 + n], n, h[c0:c0 + n])
            b["logits"][:1].copy_(logits[N - 1:N])
        check(L.samd_argmax_rows(_ptr(b["logits"]), dt, 1, s.vocab, s.vocab, None, _ptr(b["argmax"]), st))
        session.set_cache_length(N)
        session.set_start_token(b["argmax"])
        return b["logits"][0]

    def warm(self, R):
        """run the launch sequence of bucket R once with n = 0 rows (no K/V row is written, every query row is
        masked): creates the GEMM library's handles/workspaces before hipGraph capture without touching the request."""
        self.pf_n.zero_()
        scratch_L = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.forward_rows(R, self.pf_tokens, self.pf_relpos, self.pf_mask, scratch_L, self.pf_n)
        torch.cuda.current_stream().synchronize()

    def hidden_rows(self, R):
        """last hidden states (after the final norm) of the most recent forward of bucket R: what the reference's patched
        LlamaForCausalLM.forward returns as `last_hidden_states` (SO/model_patch/llama.py:112-202)."""
        return self._buffers(R)["h"]

    def forward_tokens(self, session: Session, tokens, relpos, mask_rows, n, L, return_hidden=False):
        """granular verify (SamdModel.decode): explicit draft tokens / relative positions / u64 mask rows -> logits [n, V]."""
        R = self.bucket(n)
        tok = torch.zeros(MAX_DRAFT, dtype=torch.int32, device=self.device)
        tok[:n] = tokens.reshape(-1)[:n].to(torch.int32)
        rel = torch.zeros(MAX_DRAFT, dtype=torch.int32, device=self.device)
        rel[:n] = relpos.reshape(-1)[:n].to(torch.int32)
        d_n = torch.tensor([n], dtype=torch.int32, device=self.device)
        session.set_cache_length(L)
        b = self.forward_rows(R, tok, rel, mask_rows, session.device_views()["cache_length"], d_n)
        torch.cuda.current_stream().synchronize()
        return (b["logits"][:n], b["h"][:n]) if return_hidden else b["logits"][:n]

    def verify(self, session: Session, R):
        """SamdModel.decode's LM call (SO/samd_model.py:134-138) on the session's current draft."""
        v = session.device_views()
        n_ptr = C.c_void_p(v["dmeta"] + 4)          # dmeta[D_N]
        return self.forward_rows(R, v["tokens"], v["position"], v["mask"], v["cache_length"], n_ptr)

    def compact(self, session: Session):
        """SamdStaticCache.select_indices (SO/cache.py:118-133) for all 2 x layers tensors in one launch."""
        s = self.shape
        session.kv_compact(self.kv_ptrs, 2 * s.layers, s.kv_heads, self.max_len, s.head_dim, self.kv.element_size(), n_transposed=s.layers if self.v_transposed else 0)
