"""GPU parity of the weight-streaming skinny GEMM (samd_gemm_skinny) against a plain PyTorch fp32 reference of the same
product.  Tolerance: fp16/bf16 inputs, fp32 accumulation -> the only rounding is the final cast (or none for fp32
split-K partials): |err| <= 2^-10 (fp16) / 2^-7 (bf16) relative to the row's magnitude."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip


def run(A, W, rows_pad, splits, dtype):
    N, K = W.shape
    L = samd_hip.lib()
    Wrow, W = W, torch.empty_like(W)                   # the kernel reads the packed layout
    samd_hip.check(L.samd_gemm_pack_weights(samd_hip._ptr(Wrow), samd_hip._ptr(W), N, K, samd_hip.current_stream()))
    if splits == 1:
        out = torch.full((rows_pad, N), float("nan"), device="cuda", dtype=dtype)
        samd_hip.check(L.samd_gemm_skinny(samd_hip._ptr(A), samd_hip._ptr(W), rows_pad, N, K, 1, None, samd_hip._ptr(out),
                                          samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
        return out.float()
    part = torch.full((splits, rows_pad, N), float("nan"), device="cuda", dtype=torch.float32)
    samd_hip.check(L.samd_gemm_skinny(samd_hip._ptr(A), samd_hip._ptr(W), rows_pad, N, K, splits, samd_hip._ptr(part), None,
                                      samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
    return part.sum(0)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("rows_pad,N,K,splits", [(16, 128, 256, 1), (16, 4096, 4096, 4), (32, 1024, 768, 3), (64, 12288, 4096, 2), (48, 12288, 4096, 2), (48, 4096, 11008, 8), (48, 256, 512, 1),
                                                 (64, 4096, 11008, 8), (64, 32000, 4096, 1), (32, 22016, 4096, 1), (16, 128, 2816, 11)])
def test_gemm_matches_fp32_reference(dtype, tol, rows_pad, N, K, splits):
    g = torch.Generator(device="cuda").manual_seed(N + K + rows_pad)
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    W = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(dtype)
    got = run(A, W, rows_pad, splits, dtype)
    want = A.float() @ W.float().t()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows_pad,N,K,splits", [(16, 128, 256, 1), (32, 4096, 4096, 8), (48, 4096, 11008, 8), (64, 4096, 11008, 8), (64, 4096, 4096, 8), (64, 1024, 768, 3),
                                                 (16, 4096, 4096, 4), (64, 256, 512, 1)])
def test_gemm_over_the_group_major_copy_is_bit_identical(dtype, rows_pad, N, K, splits):
    """samd_gemm_skinny_groups (round 6): the split-K product of the 32 / 48 / 64-row buckets streamed from the GROUP-MAJOR copy of o_proj /
    down_proj that the norm-fold forward keeps anyway -- every output (or fp32 partial) equal bit for bit to samd_gemm_skinny over the
    128-column-tile copy, which the runner no longer makes."""
    L = samd_hip.lib()
    g = torch.Generator(device="cuda").manual_seed(3 * N + K + rows_pad)
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    W = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(dtype)
    Wt, Wg = torch.empty_like(W), torch.empty_like(W)
    samd_hip.check(L.samd_gemm_pack_weights(samd_hip._ptr(W), samd_hip._ptr(Wt), N, K, samd_hip.current_stream()))
    samd_hip.check(L.samd_gemm_pack_groups(samd_hip._ptr(W), samd_hip._ptr(Wg), N, K, samd_hip.current_stream()))
    dt = samd_hip.torch_dtype_code(dtype)
    res = []
    for fn, w in ((L.samd_gemm_skinny, Wt), (L.samd_gemm_skinny_groups, Wg)):
        if splits == 1:
            out = torch.full((rows_pad, N), float("nan"), device="cuda", dtype=dtype)
            samd_hip.check(fn(samd_hip._ptr(A), samd_hip._ptr(w), rows_pad, N, K, 1, None, samd_hip._ptr(out), dt, samd_hip.current_stream()))
        else:
            out = torch.full((splits, rows_pad, N), float("nan"), device="cuda", dtype=torch.float32)
            samd_hip.check(fn(samd_hip._ptr(A), samd_hip._ptr(w), rows_pad, N, K, splits, samd_hip._ptr(out), None, dt, samd_hip.current_stream()))
        torch.cuda.synchronize()
        res.append(out)
    assert torch.isfinite(res[1].float()).all() and torch.equal(res[0], res[1])
    assert L.samd_gemm_skinny_groups(samd_hip._ptr(A), samd_hip._ptr(Wg), 24, N, K, 1, None, samd_hip._ptr(res[1]), dt, None) == -1      # SAMD_E_INVALID


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("rows_pad,inter,K", [(16, 64, 256), (16, 11008, 4096), (32, 1408, 768), (64, 2816, 1024), (48, 11008, 4096), (48, 192, 256)])
def test_gemm_silu_epilogue_matches_reference(dtype, tol, rows_pad, inter, K):
    """samd_gemm_skinny_silu = LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) with the model dtype's roundings."""
    g = torch.Generator(device="cuda").manual_seed(inter + K + rows_pad)
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    Wg = (torch.randn((inter, K), generator=g, device="cuda") * 0.05).to(dtype)
    Wu = (torch.randn((inter, K), generator=g, device="cuda") * 0.05).to(dtype)
    L = samd_hip.lib()
    inter_leaved = torch.stack([Wg.view(inter // 64, 64, K), Wu.view(inter // 64, 64, K)], dim=1).reshape(2 * inter, K).contiguous()
    packed = torch.empty_like(inter_leaved)
    samd_hip.check(L.samd_gemm_pack_weights(samd_hip._ptr(inter_leaved), samd_hip._ptr(packed), 2 * inter, K, samd_hip.current_stream()))
    out = torch.full((rows_pad, inter), float("nan"), device="cuda", dtype=dtype)
    samd_hip.check(L.samd_gemm_skinny_silu(samd_hip._ptr(A), samd_hip._ptr(packed), rows_pad, 2 * inter, K, samd_hip._ptr(out),
                                           samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
    gate = (A.float() @ Wg.float().t()).to(dtype)
    up = (A.float() @ Wu.float().t()).to(dtype)
    want = (torch.nn.functional.silu(gate.float()).to(dtype).float() * up.float())
    assert torch.isfinite(out).all()
    err = (out.float() - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("seed", range(6))
def test_gemm_random_chunk_and_split_counts(seed):
    """the pipeline has separate paths for 1, 2 and >= 3 chunks per workgroup and for odd / even counts: sweep them all
    (K = 256 x 1..20, every split count that divides the work unevenly too), all three row tiles, both dtypes."""
    rng = np.random.default_rng(seed)
    for _ in range(10):
        chunks = int(rng.integers(1, 21))
        K, N = 256 * chunks, 128 * int(rng.integers(1, 9))
        splits = int(rng.integers(1, chunks + 1))
        rows_pad = int(rng.choice([16, 32, 48, 64]))
        dtype, tol = ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2))[int(rng.integers(0, 2))]
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
        A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
        W = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(dtype)
        got = run(A, W, rows_pad, splits, dtype)
        want = A.float() @ W.float().t()
        err = (got - want).abs().max().item()
        assert torch.isfinite(got).all() and err <= tol * max(1.0, want.abs().max().item()), (rows_pad, N, K, splits, dtype, err)


def test_pack_weights_is_the_documented_permutation():
    """unit (2b + j) * 512 + tid of block (t, c) holds W[128 t + 16 w + n][256 c + 64 b + 16 g + 8 j ..+7], tid = 64 w + 16 g + n
    (include/samd_hip.h, gemm_kernels.hip header)."""
    N, K = 256, 768
    W = torch.arange(N * K, device="cuda", dtype=torch.int32).to(torch.int16).view(N, K)      # any 2-byte payload
    out = torch.empty_like(W)
    samd_hip.check(samd_hip.lib().samd_gemm_pack_weights(samd_hip._ptr(W), samd_hip._ptr(out), N, K, samd_hip.current_stream()))
    want = W.view(N // 128, 8, 16, K // 256, 4, 4, 2, 8).permute(0, 3, 4, 6, 1, 5, 2, 7).contiguous()     # [t][c][b][j][w][g][n][e]
    assert torch.equal(out.view(-1), want.view(-1))
    assert samd_hip.lib().samd_gemm_pack_weights(samd_hip._ptr(W), samd_hip._ptr(W), N, K, None) != 0           # in place: refused


def test_gemm_rejects_bad_shapes():
    L = samd_hip.lib()
    a = torch.zeros((16, 256), device="cuda", dtype=torch.float16)
    w = torch.zeros((128, 256), device="cuda", dtype=torch.float16)
    o = torch.zeros((16, 128), device="cuda", dtype=torch.float16)
    ok = lambda *args: L.samd_gemm_skinny(samd_hip._ptr(a), samd_hip._ptr(w), *args, None, samd_hip._ptr(o), samd_hip.F16, None)
    assert ok(16, 128, 256, 1) == 0
    assert ok(8, 128, 256, 1) != 0         # rows_pad must be 16 / 32 / 48 / 64
    assert ok(16, 100, 256, 1) != 0        # N % 128
    assert ok(16, 128, 200, 1) != 0        # K % 256
    assert ok(16, 128, 256, 2) != 0        # more splits than chunks
    assert L.samd_gemm_splits(4096, 4096, 16) == 8 and L.samd_gemm_splits(4096, 4096, 64) == 8
    assert L.samd_gemm_splits(22016, 4096, 16) == 1 and L.samd_gemm_splits(12288, 4096, 16) == 2      # one balanced wave of workgroups


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("rows_pad,H,Hkv,K,n,L", [(16, 32, 32, 4096, 11, 700), (64, 32, 32, 4096, 60, 1900), (32, 2, 2, 512, 32, 0), (48, 4, 1, 768, 37, 129),
                                                  (16, 3, 3, 256, 1, 5), (16, 8, 1, 512, 9, 40), (32, 40, 40, 256, 20, 3), (64, 5, 5, 512, 64, 77)])
def test_qkv_projection_with_rope_epilogue(dtype, tol, rows_pad, H, Hkv, K, n, L):
    """samd_gemm_qkv_rope (q|k|v projection + RoPE + K/V row write in one launch, no split-K; tiles of 24 or 32 rotate_half pairs that may
    straddle heads and the q / k / v regions: 48-column tiles for the 7B shape and the small ones, 64-column tiles where 48 does not divide
    the matrix -- 8 + 2 x 1 heads -- or would need a second round of workgroups -- 3 x 40 heads)
    against fp32 arithmetic, and against the two-launch path it replaces (samd_gemm_skinny + samd_rope_kv_write_cs): q rows < n, K / V
    rows [L, L + n) of the caches; nothing else may be written."""
    Lb = samd_hip.lib()
    D, max_len = 128, 2048
    g = torch.Generator(device="cuda").manual_seed(rows_pad + H + K)
    N = (H + 2 * Hkv) * D
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    W = (torch.randn((N, K), generator=g, device="cuda") * (K ** -0.5)).to(dtype)
    W64 = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_qkv64(samd_hip._ptr(W), samd_hip._ptr(W64), H + 2 * Hkv, K, samd_hip.current_stream()))
    pos = torch.randint(0, 64, (rows_pad,), generator=g, device="cuda")
    ang = (L + pos)[:, None].double() * (1.0 / (10000.0 ** (torch.arange(0, D, 2, device="cuda").double() / D)))[None, :]
    cs = torch.zeros((64, D), dtype=torch.float32, device="cuda")
    cs[:rows_pad, :64] = ang.cos().float(); cs[:rows_pad, 64:] = ang.sin().float()
    d_L = torch.tensor([L], dtype=torch.int32, device="cuda")
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    POISON = 7.0
    q = torch.full((rows_pad, H, D), POISON, device="cuda", dtype=dtype)
    kc = torch.full((Hkv, max_len, D), POISON, device="cuda", dtype=dtype)
    vc = torch.full((Hkv, max_len, D), POISON, device="cuda", dtype=dtype)
    samd_hip.check(Lb.samd_gemm_qkv_rope(samd_hip._ptr(A), samd_hip._ptr(W64), rows_pad, K, samd_hip._ptr(cs), samd_hip._ptr(d_L), samd_hip._ptr(d_n),
                                         samd_hip._ptr(q), samd_hip._ptr(kc), samd_hip._ptr(vc), H, Hkv, D, max_len, samd_hip.torch_dtype_code(dtype),
                                         samd_hip.current_stream()))
    torch.cuda.synchronize()
    # fp32 reference
    y = (A.float() @ W.float().t()).to(dtype).float().view(rows_pad, H + 2 * Hkv, D)
    c, s = cs[:rows_pad, None, :64], cs[:rows_pad, None, 64:]
    rot = torch.cat((y[..., :64] * c - y[..., 64:] * s, y[..., 64:] * c + y[..., :64] * s), dim=-1)
    scale = max(1.0, y.abs().max().item())
    assert (q[:n].float() - rot[:n, :H]).abs().max().item() <= tol * scale
    assert (kc[:, L:L + n].float() - rot[:n, H:H + Hkv].transpose(0, 1)).abs().max().item() <= tol * scale
    assert (vc[:, L:L + n].float() - y[:n, H + Hkv:].transpose(0, 1)).abs().max().item() <= tol * scale
    # nothing else was touched
    assert (q[n:] == POISON).all() and (kc[:, :L] == POISON).all() and (kc[:, L + n:] == POISON).all()
    assert (vc[:, :L] == POISON).all() and (vc[:, L + n:] == POISON).all()
    # the two-launch path on the same inputs (split-K partial sums of samd_gemm_skinny, rotated by samd_rope_kv_write_cs)
    Wp = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_weights(samd_hip._ptr(W), samd_hip._ptr(Wp), N, K, samd_hip.current_stream()))
    sp = Lb.samd_gemm_splits(N, K, rows_pad)
    part = torch.zeros((max(sp, 1), rows_pad, N), device="cuda", dtype=torch.float32)
    out = torch.zeros((rows_pad, N), device="cuda", dtype=dtype)
    samd_hip.check(Lb.samd_gemm_skinny(samd_hip._ptr(A), samd_hip._ptr(Wp), rows_pad, N, K, sp, samd_hip._ptr(part), samd_hip._ptr(out),
                                       samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
    q2, kc2, vc2 = torch.zeros_like(q), torch.zeros_like(kc), torch.zeros_like(vc)
    rel = torch.zeros(64, dtype=torch.int32, device="cuda")
    samd_hip.check(Lb.samd_rope_kv_write_cs(samd_hip._ptr(part if sp > 1 else out), samd_hip._ptr(rel), samd_hip._ptr(d_L), samd_hip._ptr(d_n), samd_hip._ptr(cs),
                                            samd_hip._ptr(q2), samd_hip._ptr(kc2), samd_hip._ptr(vc2), rows_pad, H, Hkv, D, max_len,
                                            samd_hip.torch_dtype_code(dtype), sp if sp > 1 else 0, rows_pad * N, samd_hip.current_stream()))
    torch.cuda.synchronize()
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    assert (q[:n].float() - q2[:n].float()).abs().max().item() <= 4 * ulp * scale          # same math, another fp32 summation order
    assert (kc[:, L:L + n].float() - kc2[:, L:L + n].float()).abs().max().item() <= 4 * ulp * scale
    assert (vc[:, L:L + n].float() - vc2[:, L:L + n].float()).abs().max().item() <= 4 * ulp * scale


def _pack_pairs(wg, wu, dtype):
    """[gate; up] rows interleaved in groups of 16 and packed group-major (samd_gemm_pack_groups)"""
    inter, K = wg.shape
    w = torch.stack([wg.view(inter // 16, 16, K), wu.view(inter // 16, 16, K)], dim=1).reshape(2 * inter, K).contiguous()
    out = torch.empty_like(w)
    samd_hip.check(samd_hip.lib().samd_gemm_pack_groups(samd_hip._ptr(w), samd_hip._ptr(out), 2 * inter, K, samd_hip.current_stream()))
    return out


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("rows_pad,inter,K", [(16, 11008, 4096), (64, 11008, 4096), (32, 14336, 4096), (48, 1408, 768), (16, 16, 256), (16, 48, 512), (32, 20480, 256)])
def test_gate_up_pairs_on_all_cus(dtype, tol, rows_pad, inter, K):
    """samd_gemm_pairs_silu (pairs of 16 gate + 16 up columns dealt out evenly over one workgroup per CU) against fp32 arithmetic with
    HF's roundings, and bit for bit against samd_gemm_skinny_silu (same products, same summation order per column)."""
    Lb = samd_hip.lib()
    g = torch.Generator(device="cuda").manual_seed(inter + K + rows_pad)
    A = torch.randn((rows_pad, K), generator=g, device="cuda").to(dtype)
    wg = (torch.randn((inter, K), generator=g, device="cuda") * (K ** -0.5)).to(dtype)
    wu = (torch.randn((inter, K), generator=g, device="cuda") * (K ** -0.5)).to(dtype)
    out = torch.full((rows_pad, inter), float("nan"), device="cuda", dtype=dtype)
    samd_hip.check(Lb.samd_gemm_pairs_silu(samd_hip._ptr(A), samd_hip._ptr(_pack_pairs(wg, wu, dtype)), rows_pad, inter, K, samd_hip._ptr(out),
                                           samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
    torch.cuda.synchronize()
    gate, up = (A.float() @ wg.float().t()).to(dtype).float(), (A.float() @ wu.float().t()).to(dtype).float()
    want = ((gate / (1 + torch.exp(-gate))).to(dtype).float() * up).to(dtype).float()
    assert torch.isfinite(out).all() and (out.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    if inter % 64 == 0:                                                        # the 128-column-tile kernel on the same matrices
        w128 = torch.stack([wg.view(inter // 64, 64, K), wu.view(inter // 64, 64, K)], dim=1).reshape(2 * inter, K).contiguous()
        p128 = torch.empty_like(w128)
        samd_hip.check(Lb.samd_gemm_pack_weights(samd_hip._ptr(w128), samd_hip._ptr(p128), 2 * inter, K, samd_hip.current_stream()))
        ref = torch.zeros_like(out)
        samd_hip.check(Lb.samd_gemm_skinny_silu(samd_hip._ptr(A), samd_hip._ptr(p128), rows_pad, 2 * inter, K, samd_hip._ptr(ref),
                                                samd_hip.torch_dtype_code(dtype), samd_hip.current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,K,inter", [(32, 4096, 11008), (2, 512, 256), (3, 256, 48), (6, 1280, 1024)])
def test_norm_fold_kernels(dtype, H, K, inter):
    """the norm-fold forward's launches against the launches they replace: samd_embed_rows_ssq (rows + per-tile sums of squares),
    samd_gemm_cs_residual (complete-sum projection + residual + sums of squares) vs fp32 arithmetic with the reference's roundings,
    samd_gemm_qkv_rope_norm / samd_gemm_pairs_silu_norm vs samd_rmsnorm followed by samd_gemm_qkv_rope / samd_gemm_pairs_silu."""
    Lb, st, dc = samd_hip.lib(), samd_hip.current_stream(), samd_hip.torch_dtype_code(dtype)
    P = samd_hip._ptr
    g = torch.Generator(device="cuda").manual_seed(H + K)
    hidden, D, max_len, n, L, eps = K, 128, 512, 13, 100, 1e-6
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device="cuda") * sc).to(dtype)
    # ---- embedding + sums of squares
    vocab = 97
    table, toks = rnd(vocab, hidden), torch.randint(0, vocab, (16,), generator=g, device="cuda", dtype=torch.int32)
    x = torch.zeros((16, hidden), device="cuda", dtype=dtype)
    ssq = torch.full((hidden // 16, 16), -1.0, device="cuda", dtype=torch.float32)
    samd_hip.check(Lb.samd_embed_rows_ssq(P(toks), P(table), P(x), P(ssq), 16, hidden, vocab, dc, st))
    assert torch.equal(x, table[toks.long()])
    want = x.float().view(16, hidden // 16, 16).pow(2).sum(-1).t()
    assert torch.allclose(ssq, want, rtol=1e-5, atol=1e-6)
    # ---- q|k|v with the input norm folded in, against rmsnorm -> q|k|v
    gamma = (1.0 + 0.1 * torch.randn(hidden, generator=g, device="cuda")).to(dtype)
    N = 3 * H * D
    W = rnd(N, K, sc=K ** -0.5); W64 = torch.empty_like(W)
    samd_hip.check(Lb.samd_gemm_pack_qkv64(P(W), P(W64), 3 * H, K, st))
    cs = torch.rand((64, D), generator=g, device="cuda")
    d_L = torch.tensor([L], dtype=torch.int32, device="cuda"); d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    outs = []
    for fold in (False, True):
        q = torch.zeros((16, H, D), device="cuda", dtype=dtype); kc = torch.zeros((H, max_len, D), device="cuda", dtype=dtype); vc = torch.zeros_like(kc)
        if fold:
            samd_hip.check(Lb.samd_gemm_qkv_rope_norm(P(x), P(ssq), P(gamma), eps, P(W64), 16, K, P(cs), P(d_L), P(d_n), P(q), P(kc), P(vc), H, H, D, max_len, dc, st))
        else:
            h = torch.zeros_like(x); xc = x.clone()
            samd_hip.check(Lb.samd_rmsnorm(P(xc), None, P(gamma), P(h), 16, hidden, eps, dc, 0, 0, st))
            samd_hip.check(Lb.samd_gemm_qkv_rope(P(h), P(W64), 16, K, P(cs), P(d_L), P(d_n), P(q), P(kc), P(vc), H, H, D, max_len, dc, st))
        outs.append((q, kc, vc))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        scale = max(1.0, b.float().abs().max().item())
        assert (a.float() - b.float()).abs().max().item() <= 4 * ulp * scale
    # rows_pad 8 (a draft of <= 8 nodes: only rows 0..7 of x are fetched): rows 0..7 of q / K / V bit-identical to the 16-row launch
    d_n8 = torch.tensor([7], dtype=torch.int32, device="cuda")
    eight = []
    for rows_pad in (16, 8):
        q = torch.zeros((16, H, D), device="cuda", dtype=dtype); kc = torch.zeros((H, max_len, D), device="cuda", dtype=dtype); vc = torch.zeros_like(kc)
        samd_hip.check(Lb.samd_gemm_qkv_rope_norm(P(x), P(ssq), P(gamma), eps, P(W64), rows_pad, K, P(cs), P(d_L), P(d_n8), P(q), P(kc), P(vc), H, H, D, max_len, dc, st))
        eight.append((q, kc, vc))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(*eight)) and eight[1][1][:, L:L + 7].abs().sum().item() > 0
    # ---- gate|up + SiLU with the post-attention norm folded in
    Wg, Wu = rnd(inter, K, sc=K ** -0.5), rnd(inter, K, sc=K ** -0.5)
    Wp = _pack_pairs(Wg, Wu, dtype)
    act = []
    for fold in (False, True):
        o = torch.zeros((16, inter), device="cuda", dtype=dtype)
        if fold:
            samd_hip.check(Lb.samd_gemm_pairs_silu_norm(P(x), P(ssq), P(gamma), eps, P(Wp), 16, inter, K, P(o), dc, st))
        else:
            h = torch.zeros_like(x); xc = x.clone()
            samd_hip.check(Lb.samd_rmsnorm(P(xc), None, P(gamma), P(h), 16, hidden, eps, dc, 0, 0, st))
            samd_hip.check(Lb.samd_gemm_pairs_silu(P(h), P(Wp), 16, inter, K, P(o), dc, st))
        act.append(o)
    torch.cuda.synchronize()
    assert (act[1].float() - act[0].float()).abs().max().item() <= 4 * ulp * max(1.0, act[0].float().abs().max().item())
    o8 = torch.full((16, inter), 7.0, device="cuda", dtype=dtype)
    samd_hip.check(Lb.samd_gemm_pairs_silu_norm(P(x), P(ssq), P(gamma), eps, P(Wp), 8, inter, K, P(o8), dc, st))
    torch.cuda.synchronize()
    assert torch.equal(o8[:8], act[1][:8]) and o8[8:].abs().sum().item() == 0         # rows 8..15: silu(0) * 0
    # ---- complete-sum projection + residual + sums of squares (down_proj shape: K2 = inter rounded to the chunk size)
    K2 = (inter // 256) * 256 or 256
    A2, W2 = rnd(16, K2), rnd(hidden, K2, sc=K2 ** -0.5)
    W2p = torch.empty_like(W2)
    samd_hip.check(Lb.samd_gemm_pack_groups(P(W2), P(W2p), hidden, K2, st))
    x0 = rnd(16, hidden)
    x1 = x0.clone(); ssq2 = torch.full((hidden // 16, 16), -1.0, device="cuda", dtype=torch.float32)
    samd_hip.check(Lb.samd_gemm_cs_residual(P(A2), P(W2p), 16, hidden, K2, P(x1), P(ssq2), dc, st))
    torch.cuda.synchronize()
    o_ref = (A2.float() @ W2.float().t()).to(dtype)
    x_ref = (x0.float() + o_ref.float()).to(dtype)
    scale = max(1.0, x_ref.float().abs().max().item())
    assert (x1.float() - x_ref.float()).abs().max().item() <= 3 * ulp * scale
    assert torch.allclose(ssq2, x1.float().view(16, hidden // 16, 16).pow(2).sum(-1).t(), rtol=1e-5, atol=1e-6)
    # the 8-row form (drafts of <= 8 nodes): rows 0..7 as above -- bit for bit, the arithmetic is the same -- rows 8..15 of x and ssq untouched
    x8 = x0.clone(); ssq8 = torch.full((hidden // 16, 16), -1.0, device="cuda", dtype=torch.float32)
    samd_hip.check(Lb.samd_gemm_cs_residual(P(A2), P(W2p), 8, hidden, K2, P(x8), P(ssq8), dc, st))
    torch.cuda.synchronize()
    assert torch.equal(x8[:8], x1[:8]) and torch.equal(x8[8:], x0[8:])
    assert torch.equal(ssq8[:, :8], ssq2[:, :8]) and bool((ssq8[:, 8:] == -1.0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_seam_experiment_hooks_match_the_serial_launches(dtype):
    """the two experiment entry points of profiles/r04_attention.md section 2b: samd_tree_attention_signal (the merge launch arrives on a
    device counter) and samd_gemm_cs_residual_early (o_proj that requests its weights at entry and polls that counter before it reads A)
    must give bit for bit what samd_tree_attention + samd_gemm_cs_residual give.  Run here in stream order (the producer first), twice, so
    that the epoch bookkeeping is exercised without two queues."""
    import math
    from samd_hip import _ptr, check, current_stream, lib, torch_dtype_code
    H, D, max_len, R, hidden, L0 = 32, 128, 512, 8, 4096, 200
    dt = torch_dtype_code(dtype)
    g = torch.Generator(device="cuda").manual_seed(3)
    kv = torch.randn((2, H, max_len, D), generator=g, device="cuda").to(dtype)
    q = torch.randn((16, H, D), generator=g, device="cuda").to(dtype)
    w_raw = (torch.randn((hidden, hidden), generator=g, device="cuda") * 0.02).to(dtype)
    wg = torch.empty_like(w_raw)
    check(lib().samd_gemm_pack_groups(_ptr(w_raw), _ptr(wg), hidden, hidden, current_stream()))
    mask = torch.tensor([(1 << (i + 1)) - 1 if i < 63 else -1 for i in range(64)], dtype=torch.int64, device="cuda")
    d_L = torch.tensor([L0], dtype=torch.int32, device="cuda"); d_n = torch.tensor([R - 1], dtype=torch.int32, device="cuda")
    ws = torch.zeros(lib().samd_tree_attention_workspace(R, H, D), dtype=torch.uint8, device="cuda")
    scale = 1.0 / math.sqrt(D)
    outs = []
    for early in (False, True):
        attn = torch.zeros((16, H, D), device="cuda", dtype=dtype)
        x = torch.zeros((16, hidden), device="cuda", dtype=dtype)
        ssq = torch.zeros((hidden // 16, 16), device="cuda", dtype=torch.float32)
        counter = torch.zeros(1, dtype=torch.int32, device="cuda")
        epoch = torch.zeros(hidden // 16, dtype=torch.int32, device="cuda")
        for rep in range(2):
            if early:
                check(lib().samd_tree_attention_signal(_ptr(q), _ptr(kv[0]), _ptr(kv[1]), _ptr(attn), dt, R, H, H, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                                       scale, _ptr(ws), ws.numel(), _ptr(counter), current_stream()))
                check(lib().samd_gemm_cs_residual_early(_ptr(attn), _ptr(wg), hidden, hidden, _ptr(x), _ptr(ssq), dt, _ptr(counter), _ptr(epoch), R * H,
                                                        current_stream()))
            else:
                check(lib().samd_tree_attention(_ptr(q), _ptr(kv[0]), _ptr(kv[1]), _ptr(attn), dt, R, H, H, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                                scale, _ptr(ws), ws.numel(), current_stream()))
                check(lib().samd_gemm_cs_residual(_ptr(attn), _ptr(wg), 8, hidden, hidden, _ptr(x), _ptr(ssq), dt, current_stream()))
        torch.cuda.synchronize()
        if early:
            assert int(counter.item()) == 2 * R * H and epoch.tolist() == [2] * (hidden // 16)
        outs.append((attn.clone(), x.clone(), ssq.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
