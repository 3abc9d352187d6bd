"""eagle_draft_bench.py -- one EAGLE-2 draft (extension of T accepted tokens + 5 tree levels + re-rank) of the device head at
Llama-3-8B shapes (bf16, GQA 32/8, vocabulary 128256), random weights: milliseconds per draft, wall clock around a synchronised
loop, with the tree logic in the library's kernels (default) and as PyTorch ops (SAMD_EAGLE_KERNELS=0), plus the pieces:
one head forward per row bucket, the fc projection, the per-level PyTorch ops.  usage: python scripts/eagle_draft_bench.py [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import bench
from samd_hip.llama import LlamaRunner
from samd.tree_model.eagle2 import Eagle2Head
from samd.tree_model.device_head import DeviceHead

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mcfg = dict(bench.LLAMA3_8B); mcfg["num_hidden_layers"] = 1
dtype = torch.bfloat16
runner = LlamaRunner.random_init(mcfg, 8192, dtype, seed=0)
tree_cfg = dict(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8, vocab_size=128256, rms_norm_eps=1e-5,
                rope_theta=500000.0, bias=True)
head = Eagle2Head(tree_cfg, dtype=dtype, device="cuda"); head.random_init(seed=3, std=0.02)
dh = DeviceHead(head, runner)
g = torch.Generator(device="cuda").manual_seed(0)
hs = torch.randn((T, 4096), generator=g, device="cuda").to(dtype); ids = torch.randint(3, 128256, (T + 1,), generator=g, device="cuda")

def timed(fn, iters=30):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3

def draft():
    dh.reset()
    for _ in range(20):                       # a request: L grows to ~60 + 20 T
        dh.eagle2_draft(head, hs, ids)
only = "--only-draft" in sys.argv                   # under rocprofv3: nothing but kernel-mode drafts after the set-up
for mode in (("1",) if only else ("1", "0")):
    os.environ["SAMD_EAGLE_KERNELS"] = mode
    print(f"draft (T = {T}), tree-logic kernels = {mode}: {timed(draft, 5) / 20:.3f} ms", flush=True)
os.environ["SAMD_EAGLE_KERNELS"] = "1"
if only:
    sys.exit(0)
x8 = torch.randn((8, 4096), generator=g, device="cuda").to(dtype)
pos, eye = dh.level_pos[0], torch.eye(8, device="cuda")
from samd.tree_model.device_head import _mask_rows
m8 = _mask_rows(eye)
dh.reset(); dh.extend(hs, ids[1:])
print(f"one 8-row head forward (graph replay incl. staging copies): {timed(lambda: dh._forward(x8, pos, m8, level=True)):.3f} ms")
print(f"fc projection _x (8 rows): {timed(lambda: dh._x(ids[:8].clamp(max=128255).repeat(3)[:8], x8)):.3f} ms")
lg = torch.randn((8, 128256), device="cuda", dtype=dtype)
def torch_level():
    logp = torch.log_softmax(lg.float(), dim=-1); top = torch.topk(logp, 8, dim=-1); cu = top.values + top.values[:, :1]
    best = torch.topk(cu.view(-1), 8); return best.indices // 8
print(f"per-level PyTorch ops (log_softmax + top-8 of 8 x 128256 + top-8 of 64): {timed(torch_level):.3f} ms")
