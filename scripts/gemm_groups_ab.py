"""gemm_groups_ab.py -- o_proj and down_proj of 32 layers (Vicuna-7B shapes) as one hipGraph each way: samd_gemm_skinny over the 128-column-tile
copy vs samd_gemm_skinny_groups over the group-major copy (round 6), microseconds per launch pair at the 32 / 48 / 64-row tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream
from bench import hip_time_ms
L = samd_hip.lib()
layers, H, I = 32, 4096, 11008
g = torch.Generator(device="cuda").manual_seed(0)
mats = []
for _ in range(layers):
    row = []
    for (n, k) in ((H, H), (H, I)):
        w = (torch.randn((n, k), generator=g, device="cuda") * 0.02).half()
        wt, wg = torch.empty_like(w), torch.empty_like(w)
        check(L.samd_gemm_pack_weights(_ptr(w), _ptr(wt), n, k, current_stream()))
        check(L.samd_gemm_pack_groups(_ptr(w), _ptr(wg), n, k, current_stream()))
        row.append((n, k, wt, wg))
        del w
    mats.append(row)
for R in (32, 48, 64):
    a_o = torch.randn((R, H), generator=g, device="cuda").half(); a_d = torch.randn((R, I), generator=g, device="cuda").half()
    part = torch.zeros(8 * R * H, dtype=torch.float32, device="cuda")
    out = {}
    for name, fn, sel in (("tiles", L.samd_gemm_skinny, 2), ("groups", L.samd_gemm_skinny_groups, 3)):
        def run():
            for row in mats:
                for (n, k, *_), a in zip(row, (a_o, a_d)):
                    w = row[0 if k == H else 1][sel]
                    check(fn(_ptr(a), _ptr(w), R, n, k, L.samd_gemm_splits(n, k, R), _ptr(part), None, samd_hip.F16, current_stream()))
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        out[name] = min(hip_time_ms(gr.replay, 10) for _ in range(3)) * 1e3 / layers
    print(f"rows {R}: o + down per layer  tiles {out['tiles']:.2f} us   groups {out['groups']:.2f} us")
