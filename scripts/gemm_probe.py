"""gemm_probe.py -- only the weight-streaming projections of ONE decoder layer + lm_head (k_gemm_skinny at 16 rows, Vicuna-7B
shapes, packed weights), rotating over three weight sets so that nothing is served from the Infinity Cache; for rocprofv3
--stats / --pmc passes (scripts/pmc_gemm.sh).  usage: python3 scripts/gemm_probe.py [layers_worth_of_launches] [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip

L = samd_hip.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
shapes = [("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gate_up", 22016, 4096, True), ("down", 4096, 11008, False)]
sets = []
for _ in range(3):
    ws = {}
    for name, N, K, silu in shapes:
        w = (torch.randn((N, K), device="cuda") * 0.02).half()
        p = torch.empty_like(w)
        samd_hip.check(L.samd_gemm_pack_weights(samd_hip._ptr(w), samd_hip._ptr(p), N, K, samd_hip.current_stream()))
        ws[name] = p
        del w
    sets.append(ws)
A = {K: torch.randn((R, K), device="cuda").half() for K in (4096, 11008)}
part = torch.zeros(8 * R * 22016, device="cuda", dtype=torch.float32)
out = torch.zeros((R, 22016), device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
st = samd_hip.current_stream()
nbytes = 0
for i in range(reps):
    for name, N, K, silu in shapes:
        sp = 1 if silu else L.samd_gemm_splits(N, K, R)
        w = sets[i % 3][name]
        if silu:
            samd_hip.check(L.samd_gemm_skinny_silu(samd_hip._ptr(A[K]), samd_hip._ptr(w), R, N, K, samd_hip._ptr(out), samd_hip.F16, st))
        else:
            samd_hip.check(L.samd_gemm_skinny(samd_hip._ptr(A[K]), samd_hip._ptr(w), R, N, K, sp, samd_hip._ptr(part), samd_hip._ptr(out), samd_hip.F16, st))
        nbytes += N * K * 2
torch.cuda.synchronize()
print({"launches": reps * len(shapes), "weight_bytes_per_layer": nbytes // reps})
