"""ab_step.py -- the verify forward of given row buckets as one hipGraph each (Vicuna-7B shape, random weights, L = 800), ms per replay, written
against the API every round's tree has (LlamaRunner.random_init / Session.set_draft / verify): scripts/ab_trees_run.sh copies it into each
ab_trees/<name>/scripts/ and runs it there, so every tree is measured with its own package and its own library on the SAME box.
usage: python ab_step.py [rows ...] (default 7 61: the 8- and the 64-row bucket)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import bench, samd_hip
from samd_hip.llama import LlamaRunner

ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [7, 61]
runner = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(4096)
sess.reset()
graphs = []
for n in ns:
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
    graphs.append((R, g))
out = []
for rep in range(3):                                   # three passes over the buckets: the minimum is what the box can do
    for i, (R, g) in enumerate(graphs):
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        if rep == 0:
            out.append([R, ms])
        else:
            out[i][1] = min(out[i][1], ms)
print("  ".join(f"{R}-row forward {ms:.4f} ms" for R, ms in out))
