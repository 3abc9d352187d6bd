"""gate_up_bench.py -- gate|up + SiLU*up of a decoder layer: samd_gemm_skinny_silu (128-column tiles: 172 workgroups at intermediate size
11008) against samd_gemm_pairs_silu (pairs of 16-column groups dealt over one workgroup per CU).  32 layers' worth of distinct weights
per hipGraph replay.  usage: python scripts/gate_up_bench.py [inter]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream
from bench import hip_time_ms

L = samd_hip.lib()
inter = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
K, layers = 4096, 32
P128, P16 = [], []
for _ in range(layers):
    wg = (torch.randn((inter, K), device="cuda") * 0.02).half(); wu = (torch.randn((inter, K), device="cuda") * 0.02).half()
    w128 = torch.stack([wg.view(inter // 64, 64, K), wu.view(inter // 64, 64, K)], dim=1).reshape(2 * inter, K).contiguous()
    w16 = torch.stack([wg.view(inter // 16, 16, K), wu.view(inter // 16, 16, K)], dim=1).reshape(2 * inter, K).contiguous()
    a, b = torch.empty_like(w128), torch.empty_like(w16)
    check(L.samd_gemm_pack_weights(_ptr(w128), _ptr(a), 2 * inter, K, current_stream()))
    check(L.samd_gemm_pack_groups(_ptr(w16), _ptr(b), 2 * inter, K, current_stream()))
    P128.append(a); P16.append(b)
    del wg, wu, w128, w16
for R in (16, 32, 48, 64):
    A = torch.randn((R, K), device="cuda").half()
    out = torch.zeros((R, inter), device="cuda", dtype=torch.float16)
    res = {}
    for name, fn in (("128-column tiles", lambda li: L.samd_gemm_skinny_silu(_ptr(A), _ptr(P128[li]), R, 2 * inter, K, _ptr(out), samd_hip.F16, current_stream())),
                     ("pairs on all CUs", lambda li: L.samd_gemm_pairs_silu(_ptr(A), _ptr(P16[li]), R, inter, K, _ptr(out), samd_hip.F16, current_stream()))):
        def run():
            for li in range(layers):
                check(fn(li))
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        res[name] = min(hip_time_ms(g.replay, 10) for _ in range(3)) / layers * 1e3
    print(f"rows {R}, inter {inter}: " + ", ".join(f"{k} {v:.2f} us ({4 * inter * K / v / 1e6:.2f} TB/s)" for k, v in res.items()), flush=True)
