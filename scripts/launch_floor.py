"""per-kernel floor of dependent tiny launches on one stream: eager vs hipGraph replay (silu_mul on 16x256 elements)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
L = samd_hip.lib()
gu = torch.zeros((16, 512), device="cuda", dtype=torch.float16)
out = torch.zeros((16, 256), device="cuda", dtype=torch.float16)
def chain(n):
    st = samd_hip.current_stream()
    for _ in range(n):
        L.samd_silu_mul(samd_hip._ptr(gu), samd_hip._ptr(out), 16, 256, samd_hip.F16, 0, 0, st)
def timed(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N = 200
print("eager  : %.2f us per kernel" % (timed(lambda: chain(N)) * 1e3 / N))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain(N)
print("graph  : %.2f us per kernel" % (timed(g.replay) * 1e3 / N))
a = torch.zeros(4096, device="cuda")
def tchain(n):
    for _ in range(n): a.add_(1.0)
print("torch eager add_: %.2f us" % (timed(lambda: tchain(N)) * 1e3 / N))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    tchain(N)
print("torch graph add_: %.2f us" % (timed(g2.replay) * 1e3 / N))
