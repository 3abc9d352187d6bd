"""attn_ab.py -- samd_tree_attention (split launch + merge launch) alone, 32 layers with their own K/V in one hipGraph, microseconds per layer
at the 8-, 16-, 32- and 64-row buckets; for same-box A/Bs of two builds of the library (scripts/ab_lib.sh).   usage: python scripts/attn_ab.py [L ...]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, lib

Ls = [int(a) for a in sys.argv[1:] if a.isdigit()] or [800]
H, D, layers, max_len = 32, 128, 32, 2048
Hkv = 8 if "gqa" in sys.argv[1:] else H          # "gqa": Llama-3-8B geometry (4 query heads per KV head)
Lib = lib()
kv = (torch.randn((layers, 2, Hkv, max_len, D), device="cuda") * 0.5).half()
s0 = torch.cuda.Stream()
st = samd_hip.C.c_void_p(s0.cuda_stream)
mask = torch.tensor([(1 << (i + 1)) - 1 if i < 63 else -1 for i in range(64)], dtype=torch.int64, device="cuda")
for L0 in Ls:
    row = []
    for R in (8, 16, 32, 64):
        q = torch.randn((64, H, D), device="cuda").half()
        out = torch.zeros((64, H, D), device="cuda", dtype=torch.float16)
        d_L = torch.tensor([L0], dtype=torch.int32, device="cuda"); d_n = torch.tensor([R - 1], dtype=torch.int32, device="cuda")
        ws = torch.zeros(Lib.samd_tree_attention_workspace(R, H, D), dtype=torch.uint8, device="cuda")

        def run():
            for li in range(layers):
                check(Lib.samd_tree_attention(_ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(out), samd_hip.F16, R, H, Hkv, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                              1.0 / math.sqrt(D), _ptr(ws), ws.numel(), st))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0):
            run(); s0.synchronize()
            with torch.cuda.graph(g, stream=s0):
                run()
            g.replay(); s0.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                g.replay()
            s0.synchronize()
        row.append((time.perf_counter() - t0) / 40 * 1e6 / layers)
    print(f"H={H} Hkv={Hkv} L={L0}: 8 rows {row[0]:.2f} us/layer, 16 rows {row[1]:.2f}, 32 rows {row[2]:.2f}, 64 rows {row[3]:.2f}")
