"""step_ab.py -- the verify forward of the 8- and 16-row buckets as one hipGraph each (Vicuna-7B shape, L = 800), milliseconds per replay; for
same-box A/Bs of two builds of the library (scripts/ab_lib.sh).   usage: python scripts/step_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import bench, samd_hip
from samd_hip.llama import LlamaRunner

runner = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(4096)
sess.reset()
out = []
for n in (7, 13):
    toks = torch.arange(5, 5 + n, dtype=torch.int32, device="cuda"); par = torch.arange(-1, n - 1, dtype=torch.int32, device="cuda")
    sess.set_draft(toks, par, n)
    sess.set_cache_length(800)
    R = runner.bucket(n)
    runner.warm(R)
    runner.verify(sess, R); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        runner.verify(sess, R)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    out.append((R, (time.perf_counter() - t0) / 30 * 1e3))
print("  ".join(f"{R}-row forward {ms:.4f} ms" for R, ms in out))
