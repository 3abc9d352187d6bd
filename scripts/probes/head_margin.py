"""head_margin.py -- how far the device EAGLE head's two-level tree outputs are from the PyTorch head's, relative to the tolerance of
tests/test_gpu_llama.py::test_draft_head_on_device_matches_the_pytorch_head (3e-2 of the largest reference magnitude), over several seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")]
import torch
from test_gpu_llama import tiny_llama
from samd_hip.llama import LlamaRunner
from samd.tree_model.device_head import DeviceHead
from samd.tree_model.eagle2 import Eagle2Head

rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(1.0, b.float().abs().max().item())
for seed in range(6):
    lm = tiny_llama(2, seed=8)
    runner = LlamaRunner.from_hf(lm, max_cache_len=512, dtype=torch.float16)
    tree_cfg = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, vocab_size=512, rms_norm_eps=1e-5, bias=True)
    head = Eagle2Head(tree_cfg, dtype=torch.float16, device="cuda")
    head.random_init(seed=3, std=0.08)
    dh = DeviceHead(head, runner)
    g = torch.Generator(device="cuda").manual_seed(seed)
    lm_head = runner.w["lm_head"]
    past = None
    errs = []
    for T in (70, 3):
        hs = torch.randn((T, 256), generator=g, device="cuda").half()
        ids = torch.randint(3, 512, (T,), generator=g, device="cuda")
        out, past = head.forward(hs, ids, past=past)
        last, logits = dh.extend(hs, ids)
        errs += [rel(last, out[-1:]), rel(logits, torch.nn.functional.linear(out[-1:], lm_head))]
    L = past[0].shape[1]
    ids0 = torch.randint(3, 512, (8,), generator=g, device="cuda")
    h0 = torch.randn((8, 256), generator=g, device="cuda").half()
    eye = torch.eye(8, device="cuda")
    out0, past1 = head.forward(h0, ids0, past=past, position_ids=torch.full((8,), L, device="cuda"), tree_mask=eye)
    x0 = dh._x(ids0, h0)
    d_out0, d_log0 = dh.tree(x0, torch.zeros(8, dtype=torch.int32, device="cuda"), eye)
    e0 = rel(d_out0, out0)
    par = torch.tensor([2, 2, 2, 5, 5, 5, 5, 2], device="cuda")
    ids1 = torch.randint(3, 512, (8,), generator=g, device="cuda")
    m1 = torch.cat((eye[par], eye), dim=1)
    out1, _ = head.forward(out0[par], ids1, past=past1, position_ids=torch.full((8,), L + 1, device="cuda"), tree_mask=m1)
    anc = torch.zeros((16, 16), device="cuda"); anc[:8, :8] = eye; anc[8:] = m1
    x1 = torch.cat((x0, dh._x(ids1, d_out0[par].clone())), dim=0)
    d_out1, d_log1 = dh.tree(x1, torch.tensor([0] * 8 + [1] * 8, dtype=torch.int32, device="cuda"), anc)
    # the same level fed with the PyTorch head's own level-0 states: isolates the level-1 arithmetic from the carried-over difference
    x1b = torch.cat((x0, dh._x(ids1, out0[par].clone())), dim=0)
    d_out1b, _ = dh.tree(x1b, torch.tensor([0] * 8 + [1] * 8, dtype=torch.int32, device="cuda"), anc)
    print(f"seed {seed}: extend {max(errs):.4f}  level0 {e0:.4f}  level1 states {rel(d_out1[8:], out1):.4f} logits {rel(d_log1[8:], torch.nn.functional.linear(out1, lm_head)):.4f}"
          f"  level1 with the reference's level-0 states {rel(d_out1b[8:], out1):.4f}   |out1| max {out1.float().abs().max().item():.2f}")
