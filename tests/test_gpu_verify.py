"""GPU parity of the verify-side kernels: KV-cache compaction (bit-exact vs the reference semantics of
samd_sam_only/cache.py:118-133) and tree-mask attention (vs a plain PyTorch fp32 reference with the
explicit additive mask of samd_sam_only/model_patch/llama.py:82-96; fp16/bf16 tolerance below)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from oracle import sam_oracle as O
from util import random_parents


def dev(a, dtype=torch.int32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def select_indices_reference(data, start, idx, a):
    """cache.py:118-133 on one [H, max_len, D] tensor: index_select materialises, then copy_."""
    sel = data[:, [start + i for i in idx], :].clone()
    data[:, start:start + a, :] = sel
    return data


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_kv_compact(dtype):
    rng = np.random.default_rng(3)
    H, max_len, D, n_tensors = 4, 96, 128, 6
    sess = samd_hip.Session(256)
    for trial in range(12):
        n = int(rng.integers(2, 40))
        anc = random_parents(rng, n, ["bushy", "random", "chain"][trial % 3])
        tokens = rng.integers(1, 50, n).tolist()
        # node arg-max follows one random child so that several nodes get accepted
        am = []
        for i in range(n):
            kids = [j for j in range(n) if anc[j] == i]
            am.append(tokens[kids[int(rng.integers(0, len(kids)))]] if kids else 0 if trial % 4 == 0 else 77)
        start = int(rng.integers(1, 40))
        tensors = [torch.randn((H, max_len, D), device="cuda").to(dtype) for _ in range(n_tensors)]
        want = [t.clone() for t in tensors]
        ptrs = torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device="cuda")
        is_tree = trial % 5 != 4
        sess.set_draft(dev(tokens), dev(anc if is_tree else [i - 1 for i in range(n)]), n, type_=1 if is_tree else 0)
        sess.set_cache_length(start)
        sess.accept(dev(am))
        samd_hip.check(samd_hip.lib().samd_kv_compact(sess._h, samd_hip._ptr(ptrs), n_tensors, H, max_len, D, tensors[0].element_size(),
                                                      samd_hip.current_stream()))
        v = sess.read_verdict()
        a, idx = v.accept, list(v.kv_index[:v.accept])
        ret = None if not is_tree else O.gen_buffers(anc)["tree_retrieve_indices"]
        ob, oa, _ = O.eval_posterior(am, tokens, ret)
        assert (ob, oa) == (v.best, v.accept)
        if is_tree:
            assert idx == ret[ob][:oa].tolist()
            for t in want:
                select_indices_reference(t, start, idx, a)
        for t, w in zip(tensors, want):
            assert torch.equal(t, w)
        assert sess.get_cache_length() == start + a


def reference_attention(q, k_cache, v_cache, L, n, mask_rows, scale):
    """fp32 SDPA with the reference's mask: new tokens see all L cached keys and their tree ancestors."""
    H, Hkv = q.shape[1], k_cache.shape[0]
    qf = q[:n].float().permute(1, 0, 2)                                    # [H, n, D]
    kf = k_cache[:, :L + n].float().repeat_interleave(H // Hkv, dim=0)      # [H, L+n, D]
    vf = v_cache[:, :L + n].float().repeat_interleave(H // Hkv, dim=0)
    bias = torch.zeros((n, L + n), device=q.device)
    tm = torch.tensor([[(mask_rows[i] >> j) & 1 for j in range(n)] for i in range(n)], device=q.device)
    bias[:, L:] = torch.where(tm == 1, 0.0, float("-inf"))
    s = torch.einsum("hnd,hkd->hnk", qf, kf) * scale + bias
    return torch.einsum("hnk,hkd->hnd", torch.softmax(s, dim=-1), vf).permute(1, 0, 2)   # [n, H, D]


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("H,Hkv,L,n,shape", [(32, 32, 0, 1, "chain"), (32, 32, 1, 5, "chain"), (32, 32, 1000, 60, "bushy"),
                                             (32, 8, 700, 63, "random"), (4, 4, 2047 - 64, 64, "star"), (8, 2, 130, 17, "chain"),
                                             (32, 32, 511, 33, "random")])
def test_tree_attention(dtype, tol, H, Hkv, L, n, shape):
    rng = np.random.default_rng(L * 7 + n)
    D, max_len, n_pad = 128, 2048, 64
    g = torch.Generator(device="cuda").manual_seed(L + n)
    q = torch.randn((n_pad, H, D), generator=g, device="cuda").to(dtype)
    k_cache = torch.randn((Hkv, max_len, D), generator=g, device="cuda").to(dtype)
    v_cache = (torch.randn((Hkv, max_len, D), generator=g, device="cuda") * torch.linspace(0.5, 2.0, D, device="cuda")).to(dtype)
    k_cache[:, L + n:] = float("nan")          # stale rows beyond the live range must never leak
    v_cache[:, L + n:] = float("nan")
    q[n:] = float("nan")                       # padded query rows are ignored and zeroed
    anc = random_parents(rng, n, shape)
    mask_rows = [int(np.uint64(x)) for x in
                 [sum(1 << j for j in range(n) if O.gen_buffers(anc)["tree_attn_mask"][0, 0][i, j]) for i in range(n)]]
    mask = torch.tensor(np.array(mask_rows + [0] * (64 - n), dtype=np.uint64).view(np.int64), device="cuda")
    out = torch.full((n_pad, H, D), 7.0, device="cuda").to(dtype)
    ws_bytes = samd_hip.lib().samd_tree_attention_workspace(n_pad, H, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    scale = 1.0 / math.sqrt(D)
    d_L, d_n = dev([L]), dev([n])              # device-side scalars (kept alive across the launch)
    samd_hip.check(samd_hip.lib().samd_tree_attention(samd_hip._ptr(q), samd_hip._ptr(k_cache), samd_hip._ptr(v_cache), samd_hip._ptr(out),
                                                      samd_hip.torch_dtype_code(dtype), n_pad, H, Hkv, D, max_len, samd_hip._ptr(mask),
                                                      samd_hip._ptr(d_L), samd_hip._ptr(d_n), scale, samd_hip._ptr(ws), ws_bytes,
                                                      samd_hip.current_stream()))
    want = reference_attention(q, k_cache, v_cache, L, n, mask_rows, scale)
    got = out[:n].float()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err < tol * max(1.0, want.abs().max().item()), err
    assert (out[n:] == 0).all()
