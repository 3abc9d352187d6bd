"""gemm_bench.py -- samd_gemm_skinny vs torch.mm (hipBLASLt) on the verify forward's shapes; GB/s of weight bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from bench import hip_time_ms

L = samd_hip.lib()
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]
for R in (16, 32, 64):
    tot_mine = tot_ref = 0.0
    for name, N, K in shapes:
        A = torch.randn((R, K), device="cuda").half()
        Ws = [(torch.randn((N, K), device="cuda") * 0.02).half() for _ in range(6)]      # rotate > L2/MALL
        Wp = [torch.empty_like(w) for w in Ws]
        for w, p in zip(Ws, Wp):
            samd_hip.check(L.samd_gemm_pack_weights(samd_hip._ptr(w), samd_hip._ptr(p), N, K, samd_hip.current_stream()))
        out = torch.zeros((R, N), device="cuda", dtype=torch.float16)
        for S in sorted({1, L.samd_gemm_splits(N, K, R), 2, 3, 4, 5, 6, 8, 12, 16}):
            if S > K // 256:
                continue
            part = torch.zeros((S, R, N), device="cuda", dtype=torch.float32)
            i = [0]
            def mine():
                i[0] = (i[0] + 1) % len(Ws)
                samd_hip.check(L.samd_gemm_skinny(samd_hip._ptr(A), samd_hip._ptr(Wp[i[0]]), R, N, K, S, samd_hip._ptr(part),
                                                  samd_hip._ptr(out), samd_hip.F16, samd_hip.current_stream()))
            ms = hip_time_ms(mine, 30)
            tag = " <- auto" if S == L.samd_gemm_splits(N, K, R) else ""
            print(f"R={R:2d} {name:8s} N={N:5d} K={K:5d} S={S}: {ms*1e3:7.1f} us  {N*K*2/ms/1e6:7.1f} GB/s{tag}")
            if tag:
                tot_mine += ms
        def ref():
            i[0] = (i[0] + 1) % len(Ws)
            torch.mm(A, Ws[i[0]].t(), out=out)
        ms = hip_time_ms(ref, 30)
        tot_ref += ms
        print(f"R={R:2d} {name:8s} torch.mm              : {ms*1e3:7.1f} us  {N*K*2/ms/1e6:7.1f} GB/s")
    print(f"R={R}: per-layer-ish sum mine {tot_mine*1e3:.1f} us vs torch {tot_ref*1e3:.1f} us")
