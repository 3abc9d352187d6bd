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
VT = "vt" in sys.argv[1:] and hasattr(Lib, "samd_tree_attention_vt")          # "vt": also time the transposed-V variant (and compare its output bit for bit)
vt = kv[:, 1].transpose(-1, -2).contiguous() if VT else None
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
        out_t = torch.zeros_like(out)

        def run_vt():
            for li in range(layers):
                check(Lib.samd_tree_attention_vt(_ptr(q), _ptr(kv[li, 0]), _ptr(vt[li]), _ptr(out_t), samd_hip.F16, R, H, Hkv, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                                 1.0 / math.sqrt(D), _ptr(ws), ws.numel(), None, st))

        def timed(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s0):
                fn(); s0.synchronize()
                with torch.cuda.graph(g, stream=s0):
                    fn()
                g.replay(); s0.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(40):
                        g.replay()
                    s0.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 40 * 1e6 / layers)
            return best
        t = timed(run)
        if VT:
            tv = timed(run_vt)
            same = bool(torch.equal(out, out_t))
            row.append(f"{t:.2f} | vt {tv:.2f}{'' if same else ' MISMATCH ' + str((out.float() - out_t.float()).abs().max().item())}")
        else:
            row.append(f"{t:.2f}")
    print(f"H={H} Hkv={Hkv} L={L0}: 8 rows {row[0]} us/layer, 16 rows {row[1]}, 32 rows {row[2]}, 64 rows {row[3]}")
