"""EAGLE-2's device-side tree logic (csrc/eagle_kernels.hip: samd_e2_rowstats / samd_e2_select / samd_e2_finish, reference
Eagle2Model.topk_genrate, samd/tree_model/eagle2/eagle2_model.py:848-913) against plain PyTorch: per-row log-softmax + top-k, and the
whole expansion against the PyTorch form of the same loop (Eagle2Head._expand_levels) on the same head and cache state.  The recorded
reference outputs are compared in tests/test_gpu_eagle_golden.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip import _ptr, check, current_stream, lib, torch_dtype_code


def make_state(depth=5, keep=62):
    dev, f32, i32 = "cuda", torch.float32, torch.int32
    n_cand = 8 + 64 * depth
    t = dict(row_lse=torch.zeros(8, dtype=f32, device=dev), top_logp=torch.zeros(64, dtype=f32, device=dev), top_idx=torch.zeros(64, dtype=i32, device=dev),
             scores=torch.zeros(16, dtype=f32, device=dev), cs_index=torch.zeros(8, dtype=i32, device=dev),
             all_scores=torch.zeros(n_cand, dtype=f32, device=dev), all_tokens=torch.zeros(n_cand, dtype=i32, device=dev),
             parents_list=torch.zeros(1 + 8 * depth, dtype=i32, device=dev), mask_rows=torch.zeros(64, dtype=torch.int64, device=dev),
             row_src=torch.zeros(8, dtype=i32, device=dev), ids=torch.zeros(8, dtype=i32, device=dev),
             rec_top_vals=torch.zeros((1 + depth) * 64, dtype=f32, device=dev), rec_top_idx=torch.zeros((1 + depth) * 64, dtype=i32, device=dev),
             rec_best_vals=torch.zeros(depth * 8, dtype=f32, device=dev), rec_best_idx=torch.zeros(depth * 8, dtype=i32, device=dev),
             rec_final_vals=torch.zeros(keep, dtype=f32, device=dev), rec_final_idx=torch.zeros(keep, dtype=i32, device=dev))
    st = samd_hip.E2State(**{k: v.data_ptr() for k, v in t.items()})
    return t, st


@pytest.mark.parametrize("split", [True, False], ids=["split", "one-workgroup"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("vocab,rows", [(512, 8), (32000, 8), (128256, 8), (32000, 1), (1000, 3), (4097, 2), (151936, 8)])
def test_rowstats_matches_log_softmax_topk(dtype, vocab, rows, split):
    g = torch.Generator(device="cuda").manual_seed(vocab + rows)
    logits = (torch.randn((rows, vocab), generator=g, device="cuda") * 4).to(dtype)
    if vocab == 4097:
        logits[:, -1] = 30.0                                  # the best element sits alone in the last split
    t, st = make_state()
    nbytes = lib().samd_e2_rowstats_workspace(vocab)
    assert nbytes == 8 * ((vocab + 4095) // 4096) * 18 * 4
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda") if split else None
    check(lib().samd_e2_rowstats(_ptr(logits), torch_dtype_code(dtype), rows, vocab, vocab, C.byref(st), _ptr(ws) if split else None, nbytes if split else 0,
                                 current_stream()))
    torch.cuda.synchronize()
    logp = torch.log_softmax(logits.float(), dim=-1)
    # expected order: value descending, index ascending among equal values (half precision produces exact ties)
    order = torch.argsort(-logits.float(), dim=-1, stable=True)[:, :8]        # ties of the (half-precision) logits keep index order
    want_v = torch.gather(logp, 1, order)
    got_v, got_i = t["top_logp"].view(8, 8)[:rows], t["top_idx"].view(8, 8)[:rows].long()
    assert torch.equal(got_i, order)
    assert (got_v - want_v).abs().max().item() < 2e-4
    assert (t["row_lse"][:rows] - torch.logsumexp(logits.float(), dim=-1)).abs().max().item() < 2e-4


@pytest.mark.parametrize("split", [True, False], ids=["split", "one-workgroup"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_rowstats_with_masked_vocabulary_entries(dtype, split):
    """-inf logits (a masked vocabulary) add nothing to the row's log-sum-exp on either path -- also when a thread meets -inf before
    any finite value (column 0 .. 4095 masked: whole threads / whole splits see only -inf first)"""
    vocab, rows = 32000, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    logits = (torch.randn((rows, vocab), generator=g, device="cuda") * 4).to(dtype)
    logits[:, :4096] = float("-inf")
    logits[1, 5000:20000] = float("-inf")
    logits[2, ::2] = float("-inf")
    t, st = make_state()
    nbytes = lib().samd_e2_rowstats_workspace(vocab)
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda") if split else None
    check(lib().samd_e2_rowstats(_ptr(logits), torch_dtype_code(dtype), rows, vocab, vocab, C.byref(st), _ptr(ws) if split else None, nbytes if split else 0,
                                 current_stream()))
    torch.cuda.synchronize()
    order = torch.argsort(-logits.float(), dim=-1, stable=True)[:, :8]
    want_v = torch.gather(torch.log_softmax(logits.float(), dim=-1), 1, order)
    assert torch.equal(t["top_idx"].view(8, 8)[:rows].long(), order)
    assert torch.isfinite(t["row_lse"][:rows]).all()
    assert (t["top_logp"].view(8, 8)[:rows] - want_v).abs().max().item() < 2e-4
    assert (t["row_lse"][:rows] - torch.logsumexp(logits.float(), dim=-1)).abs().max().item() < 2e-4


def test_expansion_kernels_match_the_pytorch_loop():
    """same head, same accepted tokens: the kernel path (samd_e2_*) and Eagle2Head._expand_levels give the same draft"""
    from eagle_fixture_weights import call_inputs
    from test_gpu_eagle_golden import device_head_for
    from samd.tree_model.eagle2 import Eagle2Head
    seed = 335
    head, runner, dh = device_head_for(seed, Eagle2Head)
    outs = {}
    for mode in ("1", "0"):
        os.environ["SAMD_EAGLE_KERNELS"] = mode
        try:
            dh.reset()
            got = []
            for ci, T in enumerate([13, 1, 4, 2, 7]):
                hs, ids = call_inputs(seed, ci, T)
                toks, par = head.topk_generate_device(dh, torch.from_numpy(hs).cuda().half(), torch.from_numpy(ids).cuda())
                torch.cuda.synchronize()
                got.append((toks.tolist(), par.tolist()))
            outs[mode] = got
        finally:
            os.environ.pop("SAMD_EAGLE_KERNELS", None)
    for a, b in zip(outs["1"], outs["0"]):
        assert a == b
        assert a[1][0] == -1 and all(0 <= a[1][i] < i for i in range(1, len(a[1])))


def test_step_fast_path_matches_the_general_path():
    """eagle2_draft_step (accepted tokens, verify rows and bonus token read from device arrays: samd_e2_stage_extend + in-place
    extension graph) gives the same draft as update() + eagle2_draft() fed with host-gathered tensors, call after call -- also for
    -1 entries of kv_index (the reference's padding: the last draft row) and for more than 8 accepted tokens (16-row bucket)."""
    from test_gpu_eagle_golden import device_head_for
    from samd.tree_model.eagle2 import Eagle2Head
    seed = 335
    head_a, _, dh_a = device_head_for(seed, Eagle2Head)
    head_b, _, dh_b = device_head_for(seed, Eagle2Head)
    g = torch.Generator(device="cuda").manual_seed(5)
    H = dh_a.embed.shape[1]
    dh_a.reset(); dh_b.reset()
    # the first call of a request carries the whole prompt: general path on both heads
    hs0 = torch.randn((21, H), generator=g, device="cuda").half(); ids0 = torch.randint(3, 512, (22,), generator=g, device="cuda")
    assert [x.tolist() for x in dh_a.eagle2_draft(head_a, hs0, ids0)] == [x.tolist() for x in dh_b.eagle2_draft(head_b, hs0, ids0)]
    for T, n_rows in [(1, 61), (3, 61), (7, 61), (13, 20), (2, 9), (1, 1)]:
        rows = torch.randn((64, H), generator=g, device="cuda").half()                    # the verify forward's hidden rows (bucket buffer)
        kv = torch.randint(0, n_rows, (64,), generator=g, device="cuda", dtype=torch.int32)
        if T > 1:
            kv[T - 1] = -1
        acc = torch.randint(3, 512, (64,), generator=g, device="cuda", dtype=torch.int32)
        start = torch.randint(3, 512, (1,), generator=g, device="cuda", dtype=torch.int32)
        views = dict(kv_index=kv.data_ptr(), acc_tokens=acc.data_ptr(), start_token=start.data_ptr())
        assert dh_b.fast_step_ok(head_b, T)
        got = dh_b.eagle2_draft_step(head_b, rows, views, T, n_rows)
        idx = torch.where(kv[:T] < 0, torch.full_like(kv[:T], n_rows - 1), kv[:T]).long()
        want = dh_a.eagle2_draft(head_a, rows[idx], torch.cat((acc[:T].long(), start.long())))
        torch.cuda.synchronize()
        assert got[0].tolist() == want[0].tolist() and got[1].tolist() == want[1].tolist(), (T, n_rows)
        assert dh_a.length == dh_b.length and int(dh_b.L.item()) == dh_b.length
