"""The verify path over a TRANSPOSED V cache (round 6: V kept as [H_kv][D][max_len], what LlamaRunner's "split" attention now uses wherever
max_cache_len is a multiple of 8).  Every launch that writes or reads V has a `_vt` twin; each is held to its row-major original bit for
bit (same MFMAs over the same key order, same roundings), and the attention also to the fp32 restatement of the reference's masked SDPA
(SO/model_patch/llama.py:82-96):
  samd_tree_attention_vt        <= 16 rows: one wave per (head, KV split), V^T operands straight from memory; wider: V^T tile copied into LDS
  samd_gemm_qkv_rope_vt / samd_gemm_qkv_rope_norm_vt / samd_rope_kv_write_cs_vt     the new rows' V^T columns
  samd_prefill_attention_vt     the prompt's causal attention
  LlamaRunner                   SAMD_V_LAYOUT=rows against the default: same logits, same tokens, same cache rows."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip import _ptr as P
from test_gpu_verify import reference_attention
from test_gpu_wide_drafts import mask_words
from util import random_parents


def dev(a, dtype=torch.int32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def transposed(v_cache):
    """[H_kv][max_len][D] -> the same numbers as [H_kv][D][max_len] (contiguous)"""
    return v_cache.transpose(1, 2).contiguous()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H,Hkv,L,n,n_pad,max_len,shape", [
    (32, 32, 0, 1, 8, 2048, "chain"), (32, 32, 1, 5, 8, 2048, "chain"), (32, 32, 800, 8, 8, 2048, "bushy"), (32, 8, 700, 7, 8, 2048, "random"),
    (32, 32, 1000, 16, 16, 2048, "bushy"), (8, 2, 130, 13, 16, 2048, "chain"), (4, 4, 2048 - 16, 16, 16, 2048, "star"),     # the cache's last keys
    (4, 2, 184, 16, 16, 200, "random"), (4, 4, 191, 8, 8, 200, "chain"), (3, 3, 0, 3, 8, 8, "chain"),                        # max_len % 64 != 0, a cache shorter than a tile
    (32, 32, 511, 33, 48, 2048, "random"), (32, 8, 700, 63, 64, 2048, "random"), (4, 4, 2048 - 64, 64, 64, 2048, "star"), (2, 1, 150, 30, 32, 200, "bushy"),
    (32, 32, 700, 128, 128, 2048, "bushy"), (4, 2, 37, 100, 128, 2048, "random"), (8, 8, 72, 128, 128, 200, "chain")])
def test_tree_attention_over_the_transposed_cache(dtype, tol, H, Hkv, L, n, n_pad, max_len, shape):
    rng = np.random.default_rng(L * 7 + n)
    D = 128
    g = torch.Generator(device="cuda").manual_seed(L + n)
    q = torch.randn((n_pad, H, D), generator=g, device="cuda").to(dtype)
    k_cache = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    v_cache = (torch.randn((Hkv, max_len, D), generator=g, device="cuda") * torch.linspace(0.5, 2.0, D, device="cuda")).to(dtype)
    k_cache[:, L + n:] = float("nan")          # stale rows beyond the live range must never leak
    v_cache[:, L + n:] = float("nan")
    q[n:] = float("nan")
    anc = random_parents(rng, n, shape)
    rows, mask = mask_words(anc, n)            # low words of all rows, then the high words (read only at n_pad > 64)
    vt_cache = transposed(v_cache)
    lib, dc = samd_hip.lib(), samd_hip.torch_dtype_code(dtype)
    ws_bytes = lib.samd_tree_attention_workspace(n_pad, H, D)
    scale = 1.0 / math.sqrt(D)
    d_L, d_n = dev([L]), dev([n])
    outs = []
    for vt in (False, True):
        out = torch.full((n_pad, H, D), 7.0, device="cuda").to(dtype)
        ws = torch.full((ws_bytes,), 0xFF, dtype=torch.uint8, device="cuda")        # NaN partials wherever a split does not write
        if vt:
            samd_hip.check(lib.samd_tree_attention_vt(P(q), P(k_cache), P(vt_cache), P(out), dc, n_pad, H, Hkv, D, max_len, P(mask), P(d_L), P(d_n), scale,
                                                      P(ws), ws_bytes, None, samd_hip.current_stream()))
        else:
            samd_hip.check(lib.samd_tree_attention(P(q), P(k_cache), P(v_cache), P(out), dc, n_pad, H, Hkv, D, max_len, P(mask), P(d_L), P(d_n), scale,
                                                   P(ws), ws_bytes, samd_hip.current_stream()))
        torch.cuda.synchronize()
        outs.append(out)
    want = reference_attention(q, k_cache, v_cache, L, n, rows, scale)
    got = outs[1][:n].float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err
    assert (outs[1][n:] == 0).all()
    assert torch.equal(outs[0], outs[1])       # the same sums in the same order as the row-major launch


def test_transposed_attention_rejects_a_cache_length_that_is_not_a_multiple_of_8():
    lib = samd_hip.lib()
    t = torch.zeros(4096, device="cuda", dtype=torch.float16)
    i = torch.zeros(128 * 2, device="cuda", dtype=torch.int64)
    z = dev([0])
    ws = torch.zeros(lib.samd_tree_attention_workspace(8, 1, 128), dtype=torch.uint8, device="cuda")
    args = lambda max_len: (P(t), P(t), P(t), P(t), samd_hip.F16, 8, 1, 1, 128, max_len, P(i), P(z), P(z), 0.1, P(ws), ws.numel(), None, samd_hip.current_stream())
    assert lib.samd_tree_attention_vt(*args(20)) < 0 and lib.samd_tree_attention_vt(*args(4)) < 0
    assert lib.samd_tree_attention_vt(*args(16)) == 0
    pa = lambda max_len: (P(t), P(t), P(t), P(t), samd_hip.F16, 2, 0, 1, 1, 128, max_len, 0.1, samd_hip.current_stream())
    assert lib.samd_prefill_attention_vt(*pa(12)) < 0 and lib.samd_prefill_attention_vt(*pa(1 << 24)) < 0
    assert lib.samd_prefill_attention_vt(*pa(16)) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows_pad,H,Hkv,K,n,L", [(16, 32, 32, 4096, 11, 700), (64, 32, 32, 4096, 60, 1900), (32, 2, 2, 512, 32, 0), (48, 4, 1, 768, 37, 129),
                                                  (16, 3, 3, 256, 1, 5), (16, 8, 1, 512, 9, 40), (64, 5, 5, 512, 64, 2048 - 64)])
def test_projection_epilogue_and_rope_launch_write_the_transposed_columns(dtype, rows_pad, H, Hkv, K, n, L):
    """samd_gemm_qkv_rope_vt and samd_rope_kv_write_cs_vt against their row-major originals: the same q rows and K rows, V[L + i][d] at
    V^T[d][L + i], and nothing else of the V^T cache touched."""
    Lb, st, dc = samd_hip.lib(), samd_hip.current_stream(), samd_hip.torch_dtype_code(dtype)
    D, max_len = 128, 2048
    g = torch.Generator(device="cuda").manual_seed(rows_pad + H + K)
    N = (H + 2 * Hkv) * D
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    W = (torch.randn((N, K), generator=g, device="cuda") * (K ** -0.5)).to(dtype)
    W64 = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_qkv64(P(W), P(W64), H + 2 * Hkv, K, st))
    cs = torch.rand((64, D), generator=g, device="cuda")
    d_L, d_n = dev([L]), dev([n])
    POISON = 7.0
    fresh = lambda: (torch.full((rows_pad, H, D), POISON, device="cuda", dtype=dtype), torch.full((Hkv, max_len, D), POISON, device="cuda", dtype=dtype),
                     torch.full((Hkv, max_len, D), POISON, device="cuda", dtype=dtype))
    q0, k0, v0 = fresh()
    samd_hip.check(Lb.samd_gemm_qkv_rope(P(A), P(W64), rows_pad, K, P(cs), P(d_L), P(d_n), P(q0), P(k0), P(v0), H, Hkv, D, max_len, dc, st))
    q1, k1, v1 = fresh()                       # v1's storage holds [Hkv][D][max_len] from here on
    samd_hip.check(Lb.samd_gemm_qkv_rope_vt(P(A), P(W64), rows_pad, K, P(cs), P(d_L), P(d_n), P(q1), P(k1), P(v1), H, Hkv, D, max_len, dc, st))
    torch.cuda.synchronize()
    v1t = v1.view(Hkv, D, max_len)
    assert torch.equal(q0, q1) and torch.equal(k0, k1)
    assert torch.equal(v1t[:, :, L:L + n].transpose(1, 2), v0[:, L:L + n]) and v0[:, L:L + n].float().abs().sum().item() > 0
    assert (v1t[:, :, :L] == POISON).all() and (v1t[:, :, L + n:] == POISON).all()
    # the two-launch form (>= 32-row buckets of a model whose q|k|v is not packed for the fused launch): split-K partials -> RoPE + K / V^T write
    Wp = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_weights(P(W), P(Wp), N, K, st))
    sp = Lb.samd_gemm_splits(N, K, rows_pad)
    part = torch.zeros((max(sp, 1), rows_pad, N), device="cuda", dtype=torch.float32)
    out = torch.zeros((rows_pad, N), device="cuda", dtype=dtype)
    samd_hip.check(Lb.samd_gemm_skinny(P(A), P(Wp), rows_pad, N, K, sp, P(part), P(out), dc, st))
    rel = torch.zeros(64, dtype=torch.int32, device="cuda")
    res = []
    for fn in (Lb.samd_rope_kv_write_cs, Lb.samd_rope_kv_write_cs_vt):
        q2, k2, v2 = fresh()
        samd_hip.check(fn(P(part if sp > 1 else out), P(rel), P(d_L), P(d_n), P(cs), P(q2), P(k2), P(v2), rows_pad, H, Hkv, D, max_len, dc, sp if sp > 1 else 0,
                          rows_pad * N, st))
        res.append((q2, k2, v2))
    torch.cuda.synchronize()
    (qa, ka, va), (qb, kb, vb) = res
    vbt = vb.view(Hkv, D, max_len)
    assert torch.equal(qa, qb) and torch.equal(ka, kb) and torch.equal(vbt[:, :, L:L + n].transpose(1, 2), va[:, L:L + n])
    assert (vbt[:, :, :L] == POISON).all() and (vbt[:, :, L + n:] == POISON).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,K,rows_pad,n", [(32, 4096, 8, 7), (32, 4096, 16, 13), (2, 512, 16, 16), (3, 256, 8, 1)])
def test_norm_fold_projection_writes_the_transposed_columns(dtype, H, K, rows_pad, n):
    Lb, st, dc = samd_hip.lib(), samd_hip.current_stream(), samd_hip.torch_dtype_code(dtype)
    g = torch.Generator(device="cuda").manual_seed(H + K)
    D, max_len, L, eps = 128, 512, 100, 1e-6
    x = torch.randn((16, K), generator=g, device="cuda").to(dtype)
    ssq = x.float().view(16, K // 16, 16).pow(2).sum(-1).t().contiguous()
    gamma = (1.0 + 0.1 * torch.randn(K, generator=g, device="cuda")).to(dtype)
    W = (torch.randn((3 * H * D, K), generator=g, device="cuda") * K ** -0.5).to(dtype); W64 = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_qkv64(P(W), P(W64), 3 * H, K, st))
    cs = torch.rand((64, D), generator=g, device="cuda")
    d_L, d_n = dev([L]), dev([n])
    res = []
    for fn in (Lb.samd_gemm_qkv_rope_norm, Lb.samd_gemm_qkv_rope_norm_vt):
        q = torch.zeros((16, H, D), device="cuda", dtype=dtype); kc = torch.zeros((H, max_len, D), device="cuda", dtype=dtype); vc = torch.full_like(kc, 7.0)
        samd_hip.check(fn(P(x), P(ssq), P(gamma), eps, P(W64), rows_pad, K, P(cs), P(d_L), P(d_n), P(q), P(kc), P(vc), H, H, D, max_len, dc, st))
        res.append((q, kc, vc))
    torch.cuda.synchronize()
    (qa, ka, va), (qb, kb, vb) = res
    vbt = vb.view(H, D, max_len)
    assert torch.equal(qa, qb) and torch.equal(ka, kb) and torch.equal(vbt[:, :, L:L + n].transpose(1, 2), va[:, L:L + n])
    assert (vbt[:, :, :L] == 7.0).all() and (vbt[:, :, L + n:] == 7.0).all() and va[:, L:L + n].float().abs().sum().item() > 0


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("rows,pos0,H,Hkv,max_len", [(300, 0, 4, 4, 2048), (1536, 0, 8, 2, 2048), (990, 8, 8, 8, 1000), (40, 4, 4, 4, 48), (1, 2, 1, 1, 8), (129, 0, 2, 1, 136),
                                                     (1100, 0, 32, 32, 2048), (2047, 1, 2, 2, 2048)])
def test_prefill_attention_over_the_transposed_cache(dtype, tol, rows, pos0, H, Hkv, max_len):
    Lb, st, dc = samd_hip.lib(), samd_hip.current_stream(), samd_hip.torch_dtype_code(dtype)
    D, total = 128, pos0 + rows
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    q = torch.randn((rows, H, D), generator=g, device="cuda").to(dtype)
    kc = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    vc = (torch.randn((Hkv, max_len, D), generator=g, device="cuda") * torch.linspace(0.5, 2.0, D, device="cuda")).to(dtype)
    kc[:, total:] = float("nan"); vc[:, total:] = float("nan")          # nothing behind the prompt may reach a row
    vt = transposed(vc)
    scale = 1.0 / math.sqrt(D)
    o_rows, o_t = torch.full((rows, H, D), 7.0, device="cuda").to(dtype), torch.full((rows, H, D), 7.0, device="cuda").to(dtype)
    samd_hip.check(Lb.samd_prefill_attention(P(q), P(kc), P(vc), P(o_rows), dc, rows, pos0, H, Hkv, D, max_len, scale, st))
    samd_hip.check(Lb.samd_prefill_attention_vt(P(q), P(kc), P(vt), P(o_t), dc, rows, pos0, H, Hkv, D, max_len, scale, st))
    torch.cuda.synchronize()
    assert torch.isfinite(o_t.float()).all()
    assert torch.equal(o_rows, o_t)
    # fp32 causal attention of a sample of rows (the row-major launch is held to the full reference in tests/test_gpu_prefill_shaping.py)
    pick = sorted(set([0, rows - 1, rows // 2] + np.random.default_rng(rows).integers(0, rows, 6).tolist()))
    kf = kc[:, :total].float().repeat_interleave(H // Hkv, dim=0); vf = vc[:, :total].float().repeat_interleave(H // Hkv, dim=0)
    for i in pick:
        s = torch.einsum("hd,hkd->hk", q[i].float(), kf[:, :pos0 + i + 1]) * scale
        want = torch.einsum("hk,hkd->hd", torch.softmax(s, dim=-1), vf[:, :pos0 + i + 1])
        assert (o_t[i].float() - want).abs().max().item() < tol * max(1.0, want.abs().max().item())


def test_runner_over_the_transposed_cache_gives_the_row_major_runner_bit_for_bit(monkeypatch):
    """LlamaRunner's default ("split" attention, V transposed) against SAMD_V_LAYOUT=rows on the same weights: prefill logits, the verify
    forward at every row bucket (norm-fold and split-K forms), the cache rows through kv_rows(), compaction after an accept -- all identical;
    a max_cache_len that is not a multiple of 8 quietly takes the row-major form."""
    from samd_hip.llama import LlamaRunner
    from test_gpu_wide_drafts import small_llama
    monkeypatch.setenv("SAMD_QKV_FUSED", "force")
    lm, cfg = small_llama(seed=4)
    V, max_len = cfg["vocab_size"], 512
    rng = np.random.default_rng(3)
    prompt = torch.tensor([rng.integers(3, V, 150).tolist()], device="cuda")
    res = {}
    for layout in ("t", "rows"):
        monkeypatch.setenv("SAMD_V_LAYOUT", layout)
        runner = LlamaRunner.from_hf(lm, max_cache_len=max_len, dtype=torch.float16, share_weights=False)
        assert runner.v_transposed == (layout == "t")
        sess = samd_hip.Session(max_len)
        runner.kv.fill_(float("nan"))
        got = [runner.prefill(sess, prompt).float().clone()]
        Lc = prompt.numel()
        for n, shape in ((5, "chain"), (8, "bushy"), (14, "random"), (30, "random"), (47, "bushy"), (64, "chain"), (100, "random")):
            anc = random_parents(np.random.default_rng(n), n, shape)
            toks = np.random.default_rng(n + 1).integers(3, V, n)
            depth = [0] * n
            for i in range(1, n):
                depth[i] = depth[anc[i]] + 1
            rows, mask = mask_words(anc, n)
            logits = runner.forward_tokens(sess, dev(toks), dev(depth), mask, n, Lc)
            got.append(logits.float().clone())
            # accept the root's chain of first children: compaction moves their K rows / V^T columns to [Lc, Lc + a)
            path = [0]
            while True:
                kids = [j for j in range(n) if j != path[-1] and anc[j] == path[-1] and j > path[-1]]
                if not kids or len(path) >= 6:
                    break
                path.append(kids[0])
            idx = torch.zeros(samd_hip.MAX_DRAFT, dtype=torch.int32, device="cuda"); idx[:len(path)] = dev(path)
            samd_hip.check(samd_hip.lib().samd_kv_compact_indices_vt(P(runner.kv_ptrs), 2 * cfg["num_hidden_layers"], cfg["num_hidden_layers"] if runner.v_transposed else 0,
                                                                      runner.shape.kv_heads, max_len, 128, 2, Lc, P(idx), len(path), samd_hip.current_stream()))
            Lc += len(path)
            torch.cuda.synchronize()
        k, v = runner.kv_rows(Lc)
        res[layout] = (got, k.float().clone(), v.float().clone())
        del runner
        torch.cuda.empty_cache()
    for a, b in zip(res["t"][0], res["rows"][0]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.equal(res["t"][1], res["rows"][1]) and torch.equal(res["t"][2], res["rows"][2])
    monkeypatch.setenv("SAMD_V_LAYOUT", "t")
    odd = LlamaRunner.from_hf(lm, max_cache_len=515, dtype=torch.float16, share_weights=False)
    assert odd.v_transposed is False
    sess = samd_hip.Session(515)
    assert torch.equal(odd.prefill(sess, prompt).float(), res["rows"][0][0])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,rows,L,max_len", [(32, 32, 200, 0, 2048), (32, 8, 1333, 5, 2048), (8, 2, 128, 704, 2048), (4, 4, 300, 8, 304), (2, 1, 129, 3, 136)])
def test_prompt_rows_reach_the_transposed_cache_through_the_tiled_writer(dtype, H, Hkv, rows, L, max_len):
    """samd_rope_kv_write_vt from 128 rows on: q and K rows by the 16-byte-lane kernel, the V^T columns by a tiled transposition of their own
    (16-byte pieces of 8 keys; element stores where L is not a multiple of 8 and for the prompt's last, partial piece).  Held to the row-major
    samd_rope_kv_write on the same rows; rows >= n, positions outside [L, L + n) and positions past the cache stay untouched."""
    lib, D = samd_hip.lib(), 128
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    qkv = torch.randn((rows, (H + 2 * Hkv) * D), generator=g, device="cuda").to(dtype)
    ang = torch.outer(torch.arange(max_len, dtype=torch.float64), 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D)))
    cos, sin = ang.cos().float().cuda().contiguous(), ang.sin().float().cuda().contiguous()
    dc = samd_hip.torch_dtype_code(dtype)
    n = rows - 3
    rel = torch.arange(rows, dtype=torch.int32, device="cuda")
    d_L, d_n = dev([L]), dev([n])
    res = []
    for fn in (lib.samd_rope_kv_write, lib.samd_rope_kv_write_vt):
        q = torch.full((rows, H, D), 7.0, device="cuda").to(dtype)
        k = torch.full((Hkv, max_len, D), 5.0, device="cuda").to(dtype)
        v = torch.full((Hkv, max_len, D), 3.0, device="cuda").to(dtype)
        samd_hip.check(fn(P(qkv), P(rel), P(d_L), P(d_n), P(cos), P(sin), P(q), P(k), P(v), rows, H, Hkv, D, max_len, max_len, dc, 0, 0, samd_hip.current_stream()))
        torch.cuda.synchronize()
        res.append((q, k, v))
    (q0, k0, v0), (q1, k1, v1) = res
    live = min(n, max_len - L)                       # rows whose position lies inside the cache
    v1t = v1.view(Hkv, D, max_len)
    assert torch.equal(q0, q1) and torch.equal(k0, k1)
    assert torch.equal(v1t[:, :, L:L + live].transpose(1, 2), v0[:, L:L + live]) and (v0[:, L:L + live] != 3.0).any()
    assert (v1t[:, :, :L] == 3.0).all() and (v1t[:, :, L + live:] == 3.0).all()


@pytest.mark.parametrize("where", [0, 1])
def test_attention_launches_carrying_warm_up_workgroups_give_the_same_rows(where, monkeypatch):
    """SAMD_L2_WARM_KB > 0 (off by default): the attention's split launch (where = 0) or its merge (where = 1) carries extra workgroups that read the
    head of the output projection's weight stream.  Over the transposed cache the <= 16-row split launch is the one-wave kernel: the extra
    workgroups must leave every bucket's logits untouched (both layouts, the non-fold launch sequence that passes the hints)."""
    from samd_hip.llama import LlamaRunner
    from test_gpu_wide_drafts import small_llama
    monkeypatch.setenv("SAMD_QKV_FUSED", "force")
    monkeypatch.setenv("SAMD_NORM_FOLD", "0")
    lm, cfg = small_llama(seed=9)
    V, max_len = cfg["vocab_size"], 512
    prompt = torch.tensor([np.random.default_rng(5).integers(3, V, 140).tolist()], device="cuda")
    res = {}
    for layout in ("t", "rows"):
        for kb in (0, 64):
            monkeypatch.setenv("SAMD_V_LAYOUT", layout)
            monkeypatch.setenv("SAMD_L2_WARM_KB", str(kb))
            monkeypatch.setenv("SAMD_L2_WARM_WHERE", str(where))
            runner = LlamaRunner.from_hf(lm, max_cache_len=max_len, dtype=torch.float16, share_weights=False)
            assert runner.warm_kb == kb and not runner.norm_fold
            sess = samd_hip.Session(max_len)
            runner.prefill(sess, prompt)
            got = []
            for n, shape in ((6, "chain"), (15, "bushy"), (40, "random")):
                anc = random_parents(np.random.default_rng(n), n, shape)
                depth = [0] * n
                for i in range(1, n):
                    depth[i] = depth[anc[i]] + 1
                _, mask = mask_words(anc, n)
                got.append(runner.forward_tokens(sess, dev(np.random.default_rng(n + 1).integers(3, V, n)), dev(depth), mask, n, prompt.numel()).float().clone())
            res[(layout, kb)] = got
            del runner
            torch.cuda.empty_cache()
    for a, b, c, d in zip(res[("t", 0)], res[("t", 64)], res[("rows", 0)], res[("rows", 64)]):
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
