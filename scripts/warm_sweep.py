"""warm_sweep.py -- what the L2 warm-up of the next projection's weight stream (csrc/warm_device.h) buys: the verify forward of
bench.py's model as one hipGraph per row bucket, replayed with SAMD_L2_WARM_KB = 0 / 16 / ... per projection workgroup.
usage: python scripts/warm_sweep.py [kb ...]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip.llama import LlamaRunner
from bench import VICUNA_7B, hip_time_ms

# arguments: kb[:delay[:where]] ...   (delay x 64 cycles before the warm loads; where: 0 = attention splits, 1 = merge launch)
kbs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if a[0].isdigit()] or [(0,), (16,), (32,), (48,), (64,), (96,), (128,)]
rows = [16, 64] if "--all" not in sys.argv else [1, 8, 16, 32, 64]
runner = LlamaRunner.random_init(dict(VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(4096)
runner.prefill(sess, torch.randint(3, 32000, (1, 800), device="cuda"))
torch.cuda.synchronize()
v = sess.device_views()
for R in rows:
    runner.pf_n.fill_(max(1, R - 3))
    out = []
    for kb in kbs:
        runner.warm_kb, runner.warm_delay, runner.warm_where = (tuple(kb) + (0, 0))[:3]
        fn = lambda: runner.forward_rows(R, runner.pf_tokens, runner.pf_relpos, runner.pf_mask, v["cache_length"], runner.pf_n)
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        ms = min(hip_time_ms(g.replay, 10) for _ in range(3))
        out.append(f"{':'.join(map(str, kb))} {ms:.3f}")
    print(f"rows {R:>2d}, L = 800: " + " | ".join(out), flush=True)
