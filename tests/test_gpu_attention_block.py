"""samd_attention_block (RoPE + K row / V^T column write + tree-mask attention + merge of the tile partials in ONE launch, V cached
transposed; csrc/attn_kernels.hip) through the C ABI:

 * the K rows and V^T columns it writes are BIT-IDENTICAL to samd_rope_kv_write's values (SamdStaticCache.update, cache.py:103-115),
   and samd_rope_kv_write_vt / samd_kv_compact_indices_vt keep the transposed layout consistent with the row-major one;
 * its output agrees with fp32 SDPA under the reference's mask (samd_sam_only/model_patch/llama.py:82-96) within the model
   dtype's tolerance (fp16 2e-3, bf16 1.6e-2 of the output's magnitude -- the same bar as the three-launch path);
 * a visible prefix shorter than the write position (a draft head's tree level whose earlier levels stay in its cache)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip import _ptr, check, current_stream, lib, torch_dtype_code
from oracle import sam_oracle as O
from util import random_parents

D = 128
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def tables(max_pos):
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    ang = torch.outer(torch.arange(max_pos, dtype=torch.float64), inv)
    return ang.cos().float().cuda().contiguous(), ang.sin().float().cuda().contiguous()


def make_qkv(g, rows, W, dtype, n_part):
    if n_part:
        part = torch.randn((n_part, rows, W), generator=g, device="cuda", dtype=torch.float32)
        return part, part.sum(0).to(dtype), n_part, rows * W
    t = torch.randn((rows, W), generator=g, device="cuda").to(dtype)
    return t, t, 0, 0


def run_block(src, n_part, stride, rel, kc, vt, dtype, n_pad, H, Hkv, max_len, mask, L, n, cos, sin, vis=None):
    """kc [Hkv, max_len, D] row-major K; vt [Hkv, D, max_len] transposed V (both updated in place) -> out [n_pad, H, D]"""
    out = torch.full((n_pad, H, D), 3.0, device="cuda").to(dtype)
    cs = torch.zeros((64, D), dtype=torch.float32, device="cuda")
    d_L, d_n = torch.tensor([L], dtype=torch.int32, device="cuda"), torch.tensor([n], dtype=torch.int32, device="cuda")
    d_vis = None if vis is None else torch.tensor([vis], dtype=torch.int32, device="cuda")
    check(lib().samd_rope_rows(_ptr(rel), _ptr(d_vis if d_vis is not None else d_L), _ptr(cos), _ptr(sin), _ptr(cs), n_pad, D, cos.shape[0], current_stream()))
    check(lib().samd_attention_block(_ptr(src), n_part, stride, _ptr(cs), _ptr(kc), _ptr(vt), _ptr(out), torch_dtype_code(dtype), n_pad, H, Hkv, D,
                                     max_len, _ptr(mask), _ptr(d_L), _ptr(d_vis), _ptr(d_n), 1.0 / math.sqrt(D), current_stream()))
    torch.cuda.synchronize()
    return out


def sdpa_reference(q, keys, vals, vis_bits, n_vis):
    """q [n, H, D] fp32; keys / vals [Hkv, K, D] fp32; key k < n_vis visible to all, key n_vis + j visible to row i iff vis_bits[i][j]"""
    n, H = q.shape[0], q.shape[1]
    Hkv, K = keys.shape[0], keys.shape[1]
    kf, vf = keys.repeat_interleave(H // Hkv, dim=0), vals.repeat_interleave(H // Hkv, dim=0)
    bias = torch.zeros((n, K), device=q.device)
    tm = torch.tensor(vis_bits, device=q.device)
    bias[:, n_vis:] = torch.where(tm[:, :K - n_vis] == 1, 0.0, float("-inf"))
    s = torch.einsum("nhd,hkd->hnk", q, kf) / math.sqrt(D) + bias
    return torch.einsum("hnk,hkd->nhd", torch.softmax(s, dim=-1), vf)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,L,n,n_pad,n_part,shape", [
    (32, 32, 0, 1, 16, 0, "chain"), (32, 32, 1, 5, 16, 2, "chain"), (32, 32, 1000, 60, 64, 2, "bushy"), (32, 8, 700, 63, 64, 5, "random"),
    (4, 4, 2048 - 64, 64, 64, 0, "star"), (8, 2, 130, 17, 32, 0, "chain"), (32, 32, 511, 33, 64, 2, "random"), (32, 32, 64, 16, 16, 2, "bushy"),
    (32, 32, 63, 8, 8, 0, "bushy"), (32, 8, 959, 1, 1, 5, "chain"), (32, 32, 1024, 16, 16, 2, "random")])
def test_attention_block_matches_unfused_writes_and_sdpa(dtype, H, Hkv, L, n, n_pad, n_part, shape):
    rng = np.random.default_rng(L * 7 + n)
    max_len, g = 2048, torch.Generator(device="cuda").manual_seed(L + n + n_part)
    W = (H + 2 * Hkv) * D
    rows = max(n_pad, 16)
    src, qkv_t, n_p, stride = make_qkv(g, rows, W, dtype, n_part)
    cos, sin = tables(max_len)
    rel = torch.zeros(64, dtype=torch.int32, device="cuda")
    anc = random_parents(rng, n, shape)
    buf = O.gen_buffers(anc)
    rel[:n] = torch.tensor(buf["tree_position_ids"][0].astype(np.int32), device="cuda")
    m = buf["tree_attn_mask"][0, 0]
    bits = [[int(m[i, j]) for j in range(n)] for i in range(n)]
    mask_rows = [sum(1 << j for j in range(n) if m[i, j]) for i in range(n)]
    mask = torch.tensor(np.array(mask_rows + [0] * (64 - n), dtype=np.uint64).view(np.int64), device="cuda")
    kc0 = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    vc0 = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    kc0[:, L:] = float("nan")
    vc0[:, L:] = float("nan")                               # stale rows beyond the committed length must never leak
    # the three-launch path's cache writes
    kc_ref, vc_ref = kc0.clone(), vc0.clone()
    q_ref = torch.zeros((rows, H, D), dtype=dtype, device="cuda")
    d_L, d_n = torch.tensor([L], dtype=torch.int32, device="cuda"), torch.tensor([n], dtype=torch.int32, device="cuda")
    check(lib().samd_rope_kv_write(_ptr(src), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cos), _ptr(sin), _ptr(q_ref), _ptr(kc_ref), _ptr(vc_ref), rows, H, Hkv, D,
                                   max_len, max_len, torch_dtype_code(dtype), n_p, stride, current_stream()))
    kc, vt = kc0.clone(), vc0.transpose(1, 2).contiguous()                                                   # V^T [Hkv, D, max_len]
    out = run_block(src, n_p, stride, rel, kc, vt, dtype, n_pad, H, Hkv, max_len, mask, L, n, cos, sin)
    vc = vt.transpose(1, 2)
    assert torch.equal(kc[:, :L + n], kc_ref[:, :L + n]) and torch.equal(vc[:, :L + n], vc_ref[:, :L + n])      # bit-identical rows
    assert torch.isnan(kc[:, L + n:].float()).all() and torch.isnan(vc[:, L + n:].float()).all()                # nothing else touched
    want = sdpa_reference(q_ref[:n].float(), kc_ref[:, :L + n].float(), vc_ref[:, :L + n].float(), bits, L)
    got = out[:n].float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < TOL[dtype] * max(1.0, want.abs().max().item()), err
    assert (out[n:] == 0).all()
    # samd_rope_kv_write_cs (cos | sin per row from samd_rope_rows instead of the position tables): bit-identical q / K / V
    kc4, vc4, q4 = kc0.clone(), vc0.clone(), torch.zeros_like(q_ref)
    cs4 = torch.zeros((64, D), dtype=torch.float32, device="cuda")
    check(lib().samd_rope_rows(_ptr(rel), _ptr(d_L), _ptr(cos), _ptr(sin), _ptr(cs4), max(n_pad, 16) if rows >= 16 else n_pad, D, max_len, current_stream()))
    check(lib().samd_rope_kv_write_cs(_ptr(src), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cs4), _ptr(q4), _ptr(kc4), _ptr(vc4), rows, H, Hkv, D, max_len,
                                      torch_dtype_code(dtype), n_p, stride, current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(q4[:n], q_ref[:n]) and torch.equal(kc4[:, :L + n], kc_ref[:, :L + n]) and torch.equal(vc4[:, :L + n], vc_ref[:, :L + n])
    # samd_tree_attention_rope (two launches, row-major V): bit-identical K / V rows, output within the same tolerance
    kc3, vc3 = kc0.clone(), vc0.clone()
    ws3 = torch.zeros(lib().samd_tree_attention_rope_workspace(n_pad, H, D), dtype=torch.uint8, device="cuda")
    out3 = torch.full((n_pad, H, D), 3.0, device="cuda").to(dtype)
    cs3 = torch.zeros((64, D), dtype=torch.float32, device="cuda")
    check(lib().samd_rope_rows(_ptr(rel), _ptr(d_L), _ptr(cos), _ptr(sin), _ptr(cs3), n_pad, D, max_len, current_stream()))
    check(lib().samd_tree_attention_rope(_ptr(src), n_p, stride, _ptr(cs3), _ptr(kc3), _ptr(vc3), _ptr(out3), torch_dtype_code(dtype), n_pad, H, Hkv, D, max_len,
                                         _ptr(mask), _ptr(d_L), _ptr(d_n), 1.0 / math.sqrt(D), _ptr(ws3), ws3.numel(), current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(kc3[:, :L + n], kc_ref[:, :L + n]) and torch.equal(vc3[:, :L + n], vc_ref[:, :L + n])
    assert torch.isnan(kc3[:, L + n:].float()).all() and torch.isnan(vc3[:, L + n:].float()).all()
    err3 = (out3[:n].float() - want).abs().max().item()
    assert torch.isfinite(out3[:n].float()).all() and err3 < TOL[dtype] * max(1.0, want.abs().max().item()), err3
    assert (out3[n:] == 0).all()
    # samd_rope_kv_write_vt writes the same V^T columns (and the same K rows) as the block kernel
    kc2, vt2 = kc0.clone(), vc0.transpose(1, 2).contiguous()
    q2 = torch.zeros_like(q_ref)
    check(lib().samd_rope_kv_write_vt(_ptr(src), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cos), _ptr(sin), _ptr(q2), _ptr(kc2), _ptr(vt2), rows, H, Hkv, D,
                                      max_len, max_len, torch_dtype_code(dtype), n_p, stride, current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(q2, q_ref) and torch.equal(kc2[:, :L + n], kc[:, :L + n]) and torch.equal(vt2[:, :, :L + n], vt[:, :, :L + n])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_block_no_rows(dtype):
    """n = 0 (the warm-up run before hipGraph capture): no K/V row is written, every output row is zero"""
    H, Hkv, max_len, g = 32, 8, 512, torch.Generator(device="cuda").manual_seed(5)
    W = (H + 2 * Hkv) * D
    src, _, n_p, stride = make_qkv(g, 16, W, dtype, 0)
    cos, sin = tables(max_len)
    rel = torch.zeros(64, dtype=torch.int32, device="cuda")
    mask = torch.zeros(64, dtype=torch.int64, device="cuda")
    kc = torch.full((Hkv, max_len, D), 7.0, device="cuda").to(dtype)
    vt = torch.full((Hkv, D, max_len), 9.0, device="cuda").to(dtype)
    for L in (0, 200):
        out = run_block(src, n_p, stride, rel, kc, vt, dtype, 16, H, Hkv, max_len, mask, L, 0, cos, sin)
        assert (out == 0).all() and (kc == 7).all() and (vt == 9).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("vis,earlier,n", [(100, 16, 8), (777, 40, 8), (5, 0, 8), (300, 56, 8)])
def test_attention_block_visible_prefix(dtype, vis, earlier, n):
    """a draft head's tree level: `vis` accepted tokens visible to all rows, `earlier` rows of previous levels in the cache at
    [vis, vis + earlier) and n new rows at [vis + earlier, ...), all three governed by the u64 rows (bit j = key vis + j)."""
    H, Hkv, max_len = 32, 32, 1024
    rng = np.random.default_rng(vis + earlier)
    g = torch.Generator(device="cuda").manual_seed(vis)
    W = (H + 2 * Hkv) * D
    src, qkv_t, n_p, stride = make_qkv(g, 16, W, dtype, 2)
    cos, sin = tables(max_len)
    Lw, tot = vis + earlier, earlier + n
    bits = [[int(rng.random() < 0.4) for _ in range(tot)] for _ in range(n)]
    for i in range(n):
        bits[i][earlier + i] = 1                              # a row always sees itself
    mask_rows = [sum(1 << j for j in range(tot) if bits[i][j]) for i in range(n)]
    mask = torch.tensor(np.array(mask_rows + [0] * (64 - n), dtype=np.uint64).view(np.int64), device="cuda")
    rel = torch.zeros(64, dtype=torch.int32, device="cuda")
    rel[:n] = torch.tensor(rng.integers(0, 6, n).astype(np.int32), device="cuda")
    kc = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    vc = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    kc[:, Lw:] = float("nan")
    vc[:, Lw:] = float("nan")
    # expected rows: RoPE at position vis + rel (NOT write position + rel)
    x = qkv_t[:n].float().view(n, H + 2 * Hkv, D)
    pos = vis + rel[:n].long()
    c, s = torch.cat((cos[pos], cos[pos]), -1)[:, None, :], torch.cat((sin[pos], sin[pos]), -1)[:, None, :]
    roped = (x * c + torch.cat((-x[..., D // 2:], x[..., :D // 2]), -1) * s).to(dtype)
    kc_before, vc_before = kc.clone(), vc.clone()
    vt = vc.transpose(1, 2).contiguous()
    out = run_block(src, n_p, stride, rel, kc, vt, dtype, 16, H, Hkv, max_len, mask, Lw, n, cos, sin, vis=vis)
    vc = vt.transpose(1, 2)
    assert torch.equal(kc[:, :Lw], kc_before[:, :Lw]) and torch.equal(vc[:, :Lw], vc_before[:, :Lw])
    tol = 2 * TOL[dtype]
    assert (kc[:, Lw:Lw + n].float() - roped[:, H:H + Hkv].transpose(0, 1).float()).abs().max().item() < tol * 4
    keys = torch.cat((kc_before[:, :Lw], kc[:, Lw:Lw + n]), dim=1).float()
    vals = torch.cat((vc_before[:, :Lw], vc[:, Lw:Lw + n]), dim=1).float()
    want = sdpa_reference(roped[:, :H].float(), keys, vals, bits, vis)
    err = (out[:n].float() - want).abs().max().item()
    assert err < TOL[dtype] * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kv_compact_transposed_tensors(dtype):
    """samd_kv_compact_indices_vt: select_indices (cache.py:118-133) over K tensors [H, L, D] followed by V^T tensors [H, D, L]"""
    rng = np.random.default_rng(11)
    H, max_len, n_k, n_v = 4, 96, 3, 3
    for trial in range(8):
        ks0 = [torch.randn((H, max_len, D), device="cuda").to(dtype) for _ in range(n_k)]
        vs0 = [torch.randn((H, max_len, D), device="cuda").to(dtype) for _ in range(n_v)]
        ks = [k.clone() for k in ks0]
        vts = [v.transpose(1, 2).contiguous() for v in vs0]
        a = int(rng.integers(1, 40))
        idx = sorted(rng.choice(60, a, replace=False).tolist())
        if trial % 3 == 0:
            idx[0] = 0                                                  # a row that is already in place
        start = int(rng.integers(0, 30))
        ptrs = torch.tensor([t.data_ptr() for t in ks + vts], dtype=torch.int64, device="cuda")
        d_idx = torch.tensor(idx, dtype=torch.int32, device="cuda")
        check(lib().samd_kv_compact_indices_vt(_ptr(ptrs), n_k + n_v, n_v, H, max_len, D, ks[0].element_size(), start, _ptr(d_idx), a, current_stream()))
        torch.cuda.synchronize()
        # reference: index_select materialises, then copy_ -- on the logical [H, L, D] tensors
        for before, now in list(zip(ks0, ks)) + [(v, vt.transpose(1, 2)) for v, vt in zip(vs0, vts)]:
            want = before.clone()
            want[:, start:start + a] = before[:, [start + i for i in idx]]
            assert torch.equal(now, want)
