"""attn_bench.py -- the attention block of one decoder layer, x32 layers in one hipGraph (Vicuna-7B head geometry, fp16, q|k|v from
2 fp32 split-K partials as the streaming GEMM leaves them): microseconds per layer of
   split3   samd_rope_kv_write + samd_tree_attention (+ its combine launch)                                         -- three launches
   split    the same with samd_rope_kv_write_cs (cos | sin per row prepared once per forward)                        -- three launches
   split2   samd_tree_attention_rope (RoPE and the K/V row write inside the split kernel) + the slot merge           -- two launches
   block    samd_attention_block (V cached transposed; one workgroup per head and 16 rows)                          -- one launch
usage: python scripts/attn_bench.py [L]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream, lib
from bench import hip_time_ms

L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 800
H, Hkv, D, layers, max_len = 32, 32, 128, 32, 2048
W = (H + 2 * Hkv) * D
Lib = lib()
kv = torch.randn((layers, 2, Hkv, max_len, D), device="cuda").half()
cos = torch.rand((max_len, 64), device="cuda"); sin = torch.rand((max_len, 64), device="cuda")
for R in (8, 16, 32, 64):
    n = R - 3
    RP = max(R, 16)
    part = torch.randn((2, RP, W), device="cuda", dtype=torch.float32)
    q = torch.zeros((RP, H, D), device="cuda", dtype=torch.float16)
    out = torch.zeros((RP, H, D), device="cuda", dtype=torch.float16)
    rel = torch.arange(64, dtype=torch.int32, device="cuda")
    mask = torch.tensor([(1 << (i + 1)) - 1 if i < 63 else -1 for i in range(64)], dtype=torch.int64, device="cuda")
    d_L = torch.tensor([L0], dtype=torch.int32, device="cuda"); d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    ws = torch.zeros(Lib.samd_tree_attention_workspace(R, H, D), dtype=torch.uint8, device="cuda")
    cs = torch.zeros((64, D), dtype=torch.float32, device="cuda")
    scale = 1.0 / math.sqrt(D)

    def unfused():
        st = current_stream()
        for li in range(layers):
            check(Lib.samd_rope_kv_write(_ptr(part), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cos), _ptr(sin), _ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), R, H, Hkv, D,
                                         max_len, max_len, samd_hip.F16, 2, RP * W, st))
            check(Lib.samd_tree_attention(_ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(out), samd_hip.F16, R, H, Hkv, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                          scale, _ptr(ws), ws.numel(), st))

    def fused():
        st = current_stream()
        check(Lib.samd_rope_rows(_ptr(rel), _ptr(d_L), _ptr(cos), _ptr(sin), _ptr(cs), R, D, max_len, st))
        for li in range(layers):
            check(Lib.samd_attention_block(_ptr(part), 2, RP * W, _ptr(cs), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(out), samd_hip.F16, R, H, Hkv, D, max_len,
                                           _ptr(mask), _ptr(d_L), None, _ptr(d_n), scale, st))

    ws2 = torch.zeros(Lib.samd_tree_attention_rope_workspace(R, H, D), dtype=torch.uint8, device="cuda")

    def split2():
        st = current_stream()
        check(Lib.samd_rope_rows(_ptr(rel), _ptr(d_L), _ptr(cos), _ptr(sin), _ptr(cs), R, D, max_len, st))
        for li in range(layers):
            check(Lib.samd_tree_attention_rope(_ptr(part), 2, RP * W, _ptr(cs), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(out), samd_hip.F16, R, H, Hkv, D, max_len,
                                               _ptr(mask), _ptr(d_L), _ptr(d_n), scale, _ptr(ws2), ws2.numel(), st))

    def split_cs():
        st = current_stream()
        check(Lib.samd_rope_rows(_ptr(rel), _ptr(d_L), _ptr(cos), _ptr(sin), _ptr(cs), R, D, max_len, st))
        for li in range(layers):
            check(Lib.samd_rope_kv_write_cs(_ptr(part), _ptr(rel), _ptr(d_L), _ptr(d_n), _ptr(cs), _ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), R, H, Hkv, D,
                                            max_len, samd_hip.F16, 2, RP * W, st))
            check(Lib.samd_tree_attention(_ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(out), samd_hip.F16, R, H, Hkv, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                          scale, _ptr(ws), ws.numel(), st))

    res = {}
    for name, fn in (("split3", unfused), ("split", split_cs), ("split2", split2), ("block", fused)):
        ws.zero_()
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        res[name] = hip_time_ms(g.replay, 20) * 1e3 / layers
    print(f"R={R} n={n} L={L0}: " + "  ".join(f"{k} {v:.2f} us/layer" for k, v in res.items()), flush=True)
