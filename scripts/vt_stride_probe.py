"""vt_stride_probe.py -- does the stride between the rows of a transposed V cache (max_len x 2 bytes: 4 KiB at max_len 2048) matter?  The prompt's
attention (samd_prefill_attention / _vt) and the 8-row verify attention (samd_tree_attention / _vt) at several max_len, 32 layers of K/V each.
usage: python3 scripts/vt_stride_probe.py [rows]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr as P, check, lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1333
H, D, layers = 32, 128, 32
Lb = lib()
st = samd_hip.current_stream()
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e6 / layers
for max_len in (2048, 2048, 2056):
    kv = (torch.randn((layers, 2, H, max_len, D), device="cuda") * 0.5).half()
    vt = kv[:, 1].transpose(-1, -2).contiguous()
    q = torch.randn((N, H, D), device="cuda").half(); o = torch.empty_like(q)
    rows = bench(lambda: [check(Lb.samd_prefill_attention(P(q), P(kv[l, 0]), P(kv[l, 1]), P(o), samd_hip.F16, N, 0, H, H, D, max_len, 0.088, st)) for l in range(layers)])
    tr = bench(lambda: [check(Lb.samd_prefill_attention_vt(P(q), P(kv[l, 0]), P(vt[l]), P(o), samd_hip.F16, N, 0, H, H, D, max_len, 0.088, st)) for l in range(layers)])
    # verify attention at 8 rows, L = 800
    q8 = torch.randn((8, H, D), device="cuda").half(); o8 = torch.empty_like(q8)
    mask = torch.tensor([(1 << (i + 1)) - 1 if i < 63 else -1 for i in range(64)], dtype=torch.int64, device="cuda")
    d_L = torch.tensor([800], dtype=torch.int32, device="cuda"); d_n = torch.tensor([7], dtype=torch.int32, device="cuda")
    ws = torch.zeros(Lb.samd_tree_attention_workspace(8, H, D), dtype=torch.uint8, device="cuda")
    res = []
    for fn, vv in ((Lb.samd_tree_attention_warm, kv[:, 1]), (Lb.samd_tree_attention_vt, vt)):
        def run():
            for l in range(layers):
                check(fn(P(q8), P(kv[l, 0]), P(vv[l]), P(o8), samd_hip.F16, 8, H, H, D, max_len, P(mask), P(d_L), P(d_n), 0.088, P(ws), ws.numel(), None, st))
        g = torch.cuda.CUDAGraph()
        run(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            run()
        res.append(bench(lambda: g.replay(), reps=20))
    print(f"max_len {max_len}: prompt attention ({N} rows) row-major {rows:.1f} us/layer, transposed {tr:.1f} | 8-row verify attention row-major {res[0]:.2f}, transposed {res[1]:.2f}", flush=True)
    del kv, vt
