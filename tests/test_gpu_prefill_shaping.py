"""The wide prefill's two library-shaping measures (round 5, samd_hip/llama.py `_prefill_wide`; profiles/r05_prefill.md section 2): the causal
attention on the row count padded to a multiple of 128 (zero query rows and zero K / V rows behind the prompt) and a projection issued as two
row slices where the library's time steps up.  Neither may change what the prompt's forward computes: last-position logits and the K / V
rows against the plain form (one call per projection, attention on exactly N rows) and against fp32 HuggingFace."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip.llama import LlamaRunner
from test_gpu_wide_drafts import small_llama


def all_splits(max_rows=2048):
    return {k: {R: {r: True for r in (64, 128, 192, 256)} for R in range(256, max_rows, 256)} for k in ("wqkv", "wo", "wgu", "wdown")}


@pytest.mark.parametrize("N", [1025, 1100, 1281, 1536, 1999])
def test_padded_attention_and_split_projections_leave_the_prefill_unchanged(N, monkeypatch):
    lm, cfg = small_llama(seed=5)
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)
    sess = samd_hip.Session(2048)
    ids = torch.tensor([np.random.default_rng(N).integers(3, cfg["vocab_size"], N).tolist()], device="cuda")

    def run(plan, pad):
        monkeypatch.setattr(LlamaRunner, "PF_ATTN_PAD", pad)
        runner._pf_plan = plan
        runner.kv.fill_(float("nan"))                     # whatever lies behind the prompt in the cache must not reach a real row
        sess.reset()
        logits = runner.prefill(sess, ids).float().clone()
        torch.cuda.synchronize()
        return logits, torch.stack(runner.kv_rows(N), 1).float().clone()      # [layers, K | V, H_kv, N, D] whatever the V layout

    plain_logits, plain_kv = run({}, 1)
    shaped_logits, shaped_kv = run(all_splits(), 128)
    assert runner._pf_split("wgu", N) == ((N - 1) // 256) * 256 and N > runner.PF_SPLIT_MIN_ROWS
    assert torch.isfinite(shaped_logits).all() and torch.isfinite(shaped_kv).all()
    scale = max(1.0, plain_logits.abs().max().item())
    assert (shaped_logits - plain_logits).abs().max().item() < 0.01 * scale          # other tile shapes of the same products: fp16 roundings
    assert (shaped_kv - plain_kv).abs().max().item() < 0.01 * max(1.0, plain_kv.abs().max().item())
    with torch.no_grad():
        want = lm(input_ids=ids, logits_to_keep=1).logits[0, -1].float()
    err_plain, err_shaped = (plain_logits - want).abs().max().item(), (shaped_logits - want).abs().max().item()
    print(f"N={N}: |d logit| vs fp32 HF plain {err_plain:.4f} shaped {err_shaped:.4f} of {want.abs().max().item():.2f}")
    assert err_shaped < max(1.5 * err_plain, 0.02 * max(1.0, want.abs().max().item()))
    assert int(shaped_logits.argmax()) == int(plain_logits.argmax()) or (want.topk(2).values[0] - want.topk(2).values[1]).item() < 4 * err_shaped


def test_tune_prefill_measures_a_plan_and_short_prompts_never_split():
    lm, cfg = small_llama(layers=1, seed=6)
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)
    plan = runner.tune_prefill(2048)
    assert set(plan) == {"wqkv", "wo", "wgu", "wdown"}
    assert all(sorted(v) == [1024, 1280, 1536, 1792] for v in plan.values())
    assert all(isinstance(b, bool) for v in plan.values() for d in v.values() for b in d.values())
    assert set(runner.prefill_plan_summary()) == set(plan)
    runner._pf_plan = all_splits()
    assert runner._pf_split("wgu", 1024) == 0 and runner._pf_split("wgu", 300) == 0      # <= PF_SPLIT_MIN_ROWS: one code path for short prompts
    assert runner._pf_split("wgu", 1025) == 1024 and runner._pf_split("wgu", 1280) == 1024 and runner._pf_split("wgu", 1281) == 1280


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,rows,L", [(32, 32, 200, 0), (32, 8, 1333, 5), (8, 2, 128, 700)])
def test_rope_kv_write_of_the_prompt_rows_is_the_narrow_kernel_bit_for_bit(dtype, H, Hkv, rows, L):
    """samd_rope_kv_write picks k_rope_kv_wide (16-byte lanes, 8 heads per workgroup) from 128 rows on; below it the per-element kernel of
    the verify forward.  Same expressions and roundings: q rows and K / V cache rows must be identical, and rows >= n / past the cache untouched."""
    lib, D, max_len = samd_hip.lib(), 128, 2048
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    qkv = torch.randn((rows, (H + 2 * Hkv) * D), generator=g, device="cuda").to(dtype)
    ang = torch.outer(torch.arange(max_len, dtype=torch.float64), 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D)))
    cos, sin = ang.cos().float().cuda().contiguous(), ang.sin().float().cuda().contiguous()
    code = samd_hip.torch_dtype_code(dtype)
    n = rows - 3                                                   # the last three rows are beyond the prompt: not written

    def run(chunk):
        q = torch.full((rows, H, D), 7.0, device="cuda").to(dtype)
        k = torch.full((Hkv, max_len, D), 5.0, device="cuda").to(dtype)
        v = torch.full((Hkv, max_len, D), 3.0, device="cuda").to(dtype)
        for c0 in range(0, rows, chunk):
            m = min(chunk, rows - c0)
            rel = torch.arange(m, dtype=torch.int32, device="cuda")
            d_L = torch.tensor([L + c0], dtype=torch.int32, device="cuda")
            d_n = torch.tensor([max(0, min(m, n - c0))], dtype=torch.int32, device="cuda")
            samd_hip.check(lib.samd_rope_kv_write(samd_hip._ptr(qkv[c0:]), samd_hip._ptr(rel), samd_hip._ptr(d_L), samd_hip._ptr(d_n), samd_hip._ptr(cos),
                                                  samd_hip._ptr(sin), samd_hip._ptr(q[c0:]), samd_hip._ptr(k), samd_hip._ptr(v), m, H, Hkv, D, max_len, max_len,
                                                  code, 0, 0, samd_hip.current_stream()))
        torch.cuda.synchronize()
        return q, k, v

    wide, narrow = run(rows), run(100)                             # one call (>= 128 rows: the wide kernel) / chunks of 100 rows (the narrow one)
    for a, b in zip(wide, narrow):
        assert torch.equal(a, b)
    q, k, v = wide
    assert (q[n:] == 7.0).all() and (k[:, L + n:] == 5.0).all() and (v[:, L + n:] == 3.0).all() and (k[:, :L] == 5.0).all()
    x = qkv[:n].view(n, H + 2 * Hkv, D).float()
    pos = L + torch.arange(n, device="cuda")
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    x1, x2 = x[:, :H + Hkv, :64], x[:, :H + Hkv, 64:]
    want = torch.cat([(x1 * c - x2 * s).to(dtype), (x2 * c + x1 * s).to(dtype)], -1)
    assert torch.equal(q[:n], want[:, :H]) and torch.equal(k[:, L:L + n], want[:, H:].transpose(0, 1))
    assert torch.equal(v[:, L:L + n], qkv[:n].view(n, H + 2 * Hkv, D)[:, H + Hkv:].transpose(0, 1))


def test_tune_prefill_failure_leaves_single_calls(monkeypatch):
    lm, cfg = small_llama(layers=1, seed=7)
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)

    def boom(*a, **kw):
        raise RuntimeError("HIP out of memory (simulated)")
    monkeypatch.setattr(LlamaRunner, "_time_mm", boom)
    with pytest.warns(RuntimeWarning, match="tune_prefill"):
        assert runner.tune_prefill(2048) == {}
    assert runner._pf_split("wgu", 1500) == 0
    monkeypatch.setenv("SAMD_PREFILL_SPLIT", "0")
    monkeypatch.undo()
    monkeypatch.setenv("SAMD_PREFILL_SPLIT", "0")
    assert runner.tune_prefill(2048) == {} and runner.prefill_plan_summary() == {}


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("case", [(200, 32, 32, 0), (1333, 32, 8, 0), (128, 8, 8, 0), (1, 4, 2, 37), (700, 16, 16, 300), (129, 8, 1, 1900), (2048, 4, 4, 0),
                                  (96, 8, 8, 1952), (1536, 32, 32, 0), (63, 2, 1, 0), (65, 2, 2, 63),
                                  # round 6: (rows, H, Hkv, pos0, max_len) -- caches whose length is not a multiple of the 64-key tile with
                                  # the prompt ending inside the LAST, partial tile (its keys were read from a shifted tile base), and caches
                                  # shorter than one tile (64 rows were read past them)
                                  (990, 8, 8, 0, 1000), (130, 4, 2, 860, 1000), (1000, 4, 4, 0, 1000), (40, 4, 4, 0, 48), (7, 2, 2, 20, 27),
                                  (1, 2, 1, 0, 1), (200, 4, 4, 1801, 2001)], ids=lambda c: "x".join(map(str, c)))
def test_prefill_attention_against_fp32_softmax_attention(dtype, tol, case):
    """samd_prefill_attention (csrc/prefill_attn_device.h) against a plain fp32 causal attention in torch: every row, every head; the cache
    behind the prompt is NaN (it must not be read into a result), as is `out` before the launch (every row < rows must be written)."""
    rows, H, Hkv, pos0 = case[:4]
    max_len = case[4] if len(case) > 4 else 2048
    lib, D = samd_hip.lib(), 128
    g = torch.Generator(device="cuda").manual_seed(rows * 7 + H)
    q = (torch.randn((rows, H, D), generator=g, device="cuda") * 1.5).to(dtype)
    k = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    v = (torch.randn((Hkv, max_len, D), generator=g, device="cuda") * torch.linspace(0.5, 2.0, D, device="cuda")).to(dtype)
    total = pos0 + rows
    k[:, total:] = float("nan")
    v[:, total:] = float("nan")
    out = torch.full((rows, H * D), float("nan"), device="cuda").to(dtype)
    scale = 1.0 / D ** 0.5
    samd_hip.check(lib.samd_prefill_attention(samd_hip._ptr(q), samd_hip._ptr(k), samd_hip._ptr(v), samd_hip._ptr(out), samd_hip.torch_dtype_code(dtype),
                                              rows, pos0, H, Hkv, D, max_len, scale, samd_hip.current_stream()))
    torch.cuda.synchronize()
    kk = k[:, :total].float().repeat_interleave(H // Hkv, dim=0)                    # [H, total, D]
    vv = v[:, :total].float().repeat_interleave(H // Hkv, dim=0)
    s = torch.einsum("rhd,htd->hrt", q.float(), kk) * scale
    keys, pos = torch.arange(total, device="cuda")[None, None, :], (pos0 + torch.arange(rows, device="cuda"))[None, :, None]
    s = s.masked_fill(keys > pos, float("-inf"))
    want = torch.einsum("hrt,htd->rhd", torch.softmax(s, dim=-1), vv).reshape(rows, H * D)
    got = out.float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err


def test_prefill_attention_long_context_with_paired_row_blocks():
    """4096 prompt rows of a Llama-3 geometry (32 query / 8 KV heads, bf16, 8192-position cache): 32 row blocks x 32 heads > the CU count, so a
    workgroup takes a heavy and a light block; every row and head against fp32 attention, computed head by head."""
    lib, D, max_len, rows, H, Hkv = samd_hip.lib(), 128, 8192, 4096, 32, 8
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(4096)
    q = torch.randn((rows, H, D), generator=g, device="cuda").to(dtype)
    k = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    v = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    k[:, rows:] = float("nan")
    v[:, rows:] = float("nan")
    out = torch.full((rows, H * D), float("nan"), device="cuda").to(dtype)
    scale = 1.0 / D ** 0.5
    samd_hip.check(lib.samd_prefill_attention(samd_hip._ptr(q), samd_hip._ptr(k), samd_hip._ptr(v), samd_hip._ptr(out), samd_hip.torch_dtype_code(dtype),
                                              rows, 0, H, Hkv, D, max_len, scale, samd_hip.current_stream()))
    torch.cuda.synchronize()
    got = out.float().view(rows, H, D)
    assert torch.isfinite(got).all()
    causal = torch.ones((rows, rows), dtype=torch.bool, device="cuda").tril()
    worst = 0.0
    for h in range(H):
        kk, vv = k[h // (H // Hkv), :rows].float(), v[h // (H // Hkv), :rows].float()
        s = (q[:, h].float() @ kk.t()) * scale
        want = torch.softmax(s.masked_fill(~causal, float("-inf")), dim=-1) @ vv
        worst = max(worst, (got[:, h] - want).abs().max().item())
    assert worst < 1.6e-2, worst


def test_prefill_attention_rejects_what_it_cannot_do():
    lib = samd_hip.lib()
    q = torch.zeros((4, 2, 64), dtype=torch.float16, device="cuda")
    kv = torch.zeros((2, 16, 64), dtype=torch.float16, device="cuda")
    out = torch.zeros((4, 128), dtype=torch.float16, device="cuda")
    args = lambda **kw: [samd_hip._ptr(q), samd_hip._ptr(kv), samd_hip._ptr(kv), samd_hip._ptr(out), 0, kw.get("rows", 4), kw.get("pos0", 0), 2, kw.get("hkv", 2),
                         kw.get("d", 128), kw.get("max_len", 16), kw.get("scale", 0.1), samd_hip.current_stream()]
    assert lib.samd_prefill_attention(*args(d=64)) == -1                                                        # SAMD_E_INVALID
    assert lib.samd_prefill_attention(*args(rows=0)) < 0 and lib.samd_prefill_attention(*args(pos0=14)) < 0      # pos0 + rows > max_len
    assert lib.samd_prefill_attention(*args(hkv=3)) < 0 and lib.samd_prefill_attention(*args(scale=0.0)) < 0


@pytest.mark.parametrize("N", [300, 1100, 1536])
def test_prefill_with_own_attention_matches_the_sdpa_form(N, monkeypatch):
    lm, cfg = small_llama(seed=8)
    runner = LlamaRunner.from_hf(lm, max_cache_len=2048, dtype=torch.float16, share_weights=False)
    runner._pf_plan = {}
    sess = samd_hip.Session(2048)
    ids = torch.tensor([np.random.default_rng(N + 1).integers(3, cfg["vocab_size"], N).tolist()], device="cuda")
    res = {}
    for mode in ("own", "sdpa"):
        monkeypatch.setenv("SAMD_PREFILL_ATTENTION", mode)
        runner.kv.fill_(float("nan"))
        sess.reset()
        res[mode] = (runner.prefill(sess, ids).float().clone(), torch.stack(runner.kv_rows(N), 1).float().clone())
        torch.cuda.synchronize()
    with torch.no_grad():
        want = lm(input_ids=ids, logits_to_keep=1).logits[0, -1].float()
    e_own, e_sdpa = (res["own"][0] - want).abs().max().item(), (res["sdpa"][0] - want).abs().max().item()
    print(f"N={N}: |d logit| vs fp32 HF own attention {e_own:.4f}, SDPA {e_sdpa:.4f} of {want.abs().max().item():.2f}")
    assert torch.isfinite(res["own"][0]).all() and torch.isfinite(res["own"][1]).all()
    assert e_own < max(1.5 * e_sdpa, 0.02 * max(1.0, want.abs().max().item()))
    assert (res["own"][1] - res["sdpa"][1]).abs().max().item() < 0.01 * max(1.0, res["sdpa"][1].abs().max().item())
