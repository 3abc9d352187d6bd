"""qkv_rope_bench.py -- the q|k|v projection + RoPE + K/V row write of one Vicuna-7B layer: two launches (samd_gemm_skinny with split-K
partials + samd_rope_kv_write_cs) against one (samd_gemm_qkv_rope).  32 layers' worth of distinct weights per replay (nothing comes
from a cache), one hipGraph each, HIP events.  usage: python scripts/qkv_rope_bench.py   (GPU box; SAMD_QKV_DEPTH=2|3|4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream
from bench import hip_time_ms

L = samd_hip.lib()
H = Hkv = 32; D = 128; K = 4096; N = (H + 2 * Hkv) * D; max_len = 2048; layers = 32
Ws = [(torch.randn((N, K), device="cuda") * 0.02).half() for _ in range(layers)]
Wp, W64 = [torch.empty_like(w) for w in Ws], [torch.empty_like(w) for w in Ws]
for w, p, q in zip(Ws, Wp, W64):
    check(L.samd_gemm_pack_weights(_ptr(w), _ptr(p), N, K, current_stream()))
    check(L.samd_gemm_pack_qkv64(_ptr(w), _ptr(q), H + 2 * Hkv, K, current_stream()))
del Ws
kv = torch.zeros((layers, 2, Hkv, max_len, D), device="cuda", dtype=torch.float16)
cs = torch.rand((64, D), device="cuda")
rel = torch.zeros(64, dtype=torch.int32, device="cuda")
d_L = torch.tensor([800], dtype=torch.int32, device="cuda")
for R in (16, 32, 48, 64):
    A = torch.randn((R, K), device="cuda").half()
    q = torch.zeros((R, H, D), device="cuda", dtype=torch.float16)
    d_n = torch.tensor([R - 3], dtype=torch.int32, device="cuda")
    sp = L.samd_gemm_splits(N, K, R)
    part = torch.zeros((sp, R, N), device="cuda", dtype=torch.float32)
    out = torch.zeros((R, N), device="cuda", dtype=torch.float16)

    def two():
        st = current_stream()
        for li in range(layers):
            check(L.samd_gemm_skinny(_ptr(A), _ptr(Wp[li]), R, N, K, sp, _ptr(part), _ptr(out), samd_hip.F16, st))
            check(L.samd_rope_kv_write_cs(_ptr(part if sp > 1 else out), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cs), _ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]),
                                          R, H, Hkv, D, max_len, samd_hip.F16, sp if sp > 1 else 0, R * N, st))

    def one():
        st = current_stream()
        for li in range(layers):
            check(L.samd_gemm_qkv_rope(_ptr(A), _ptr(W64[li]), R, K, _ptr(cs), _ptr(d_L), _ptr(d_n), _ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), H, Hkv, D,
                                       max_len, samd_hip.F16, st))
    res = {}
    for name, fn in (("two launches", two), ("one launch", one)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        res[name] = min(hip_time_ms(g.replay, 10) for _ in range(3)) / layers * 1e3
    print(f"rows {R}: " + ", ".join(f"{k} {v:.2f} us per layer" for k, v in res.items()), flush=True)
