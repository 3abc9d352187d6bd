"""pairs_ablate.py -- the gate|up projection (samd_gemm_pairs_silu) of 32 layers as one hipGraph, per row tile; with the diagnostic library
(-DSAMD_GEMM_ABLATE, SAMD_HIP_LIB=scripts/ab/libsamd_hip_abl.so) and SAMD_GEMM_ABL=<bits> parts of the kernel are switched off (see
csrc/gemm_kernels.hip).  usage: python scripts/pairs_ablate.py [rows ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream
from bench import hip_time_ms
L = samd_hip.lib()
inter, K, layers = 11008, 4096, 32
P = []
for _ in range(layers):
    w = (torch.randn((2 * inter, K), device="cuda") * 0.02).half()
    b = torch.empty_like(w)
    check(L.samd_gemm_pack_groups(_ptr(w), _ptr(b), 2 * inter, K, current_stream()))
    P.append(b); del w
out = []
for R in [int(x) for x in sys.argv[1:]] or [16, 32, 64]:
    A = torch.randn((R, K), device="cuda").half()
    o = torch.zeros((R, inter), device="cuda", dtype=torch.float16)
    def run():
        for li in range(layers):
            check(L.samd_gemm_pairs_silu(_ptr(A), _ptr(P[li]), R, inter, K, _ptr(o), samd_hip.F16, current_stream()))
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    us = min(hip_time_ms(g.replay, 10) for _ in range(3)) / layers * 1e3
    out.append(f"rows {R}: {us:.2f} us")
print(f"ABL={os.environ.get('SAMD_GEMM_ABL', '0')}: " + ", ".join(out), flush=True)
