"""prefill_layout_ab.py -- the request start (LlamaRunner.prefill, Vicuna-7B shapes, tuned projection splits) over a row-major and a transposed V cache
in ONE process on one box: two runners over the same random weights.  usage: python3 scripts/prefill_layout_ab.py [lengths ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip.llama import LlamaRunner
from bench import VICUNA_7B
lengths = [int(x) for x in sys.argv[1:]] or [512, 800, 1100, 1333, 1536]
runners = {}
for lay in ("rows", "t"):
    os.environ["SAMD_V_LAYOUT"] = lay
    runners[lay] = LlamaRunner.random_init(dict(VICUNA_7B), 2048, torch.float16, seed=0)
    runners[lay].tune_prefill(2048)
sess = samd_hip.Session(2048)
g = torch.Generator().manual_seed(1)
for n in lengths:
    ids = torch.randint(3, 32000, (1, n), generator=g).cuda()
    best = {lay: 1e9 for lay in runners}
    for rep in range(4):
        for lay, r in runners.items():
            sess.reset(); torch.cuda.synchronize(); t = time.perf_counter()
            r.prefill(sess, ids); torch.cuda.synchronize()
            best[lay] = min(best[lay], (time.perf_counter() - t) * 1e3)
    k0, v0 = runners["rows"].kv_rows(n); k1, v1 = runners["t"].kv_rows(n)
    print(f"N={n}: row-major {best['rows']:.2f} ms, transposed {best['t']:.2f} ms; same cache rows: {bool(torch.equal(k0, k1) and torch.equal(v0, v1))}", flush=True)
