"""forward_ablation.py -- where the verify forward's time goes, without a profiler: hipGraph replays of the full
forward and of kernel subsets (Vicuna-7B shapes, random weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, current_stream
from samd_hip.llama import LlamaRunner
from bench import VICUNA_7B, hip_time_ms

L = samd_hip.lib()
native = "--torch-gemm" not in sys.argv
runner = LlamaRunner.random_init(dict(VICUNA_7B), 2048, torch.float16, seed=0, native_gemm=native)
s = runner.shape
sess = samd_hip.Session(4096)
ids = torch.randint(3, 32000, (1, 600), device="cuda")
runner.prefill(sess, ids)
torch.cuda.synchronize()
v = sess.device_views()
for R in (1, 8, 16, 32, 64):
    n = max(1, R - 3)
    runner.pf_n.fill_(n)
    b = runner._buffers(R)
    RP, part, dt = b["rows_pad"], b["part"], runner.dt

    def full():
        runner.forward_rows(R, runner.pf_tokens, runner.pf_relpos, runner.pf_mask, v["cache_length"], runner.pf_n)

    def gemm(a, w, out, final=False, wp=None):
        nn, k = w.shape
        if not runner.native_gemm:
            torch.mm(a[:R], w.t(), out=out[:R]); return
        sp = 1 if final else L.samd_gemm_splits(nn, k, RP)
        check(L.samd_gemm_skinny(_ptr(a), _ptr(wp if wp is not None else w), RP, nn, k, sp, _ptr(part), _ptr(out), dt, current_stream()))

    def gemms_only():
        for w, p in zip(runner.w["layers"], runner.wp["layers"] if runner.wp else runner.w["layers"]):
            gemm(b["h"], w["wqkv"], b["qkv"], wp=p["wqkv"] if p.get("wqkv") is not None else p["wqkv64"]); gemm(b["attn"].view(b["attn"].shape[0], -1), w["wo"], b["o"], wp=p["wo"])
            check(L.samd_gemm_pairs_silu(_ptr(b["h"]), _ptr(p["wgu"]), RP, s.inter, s.hidden, _ptr(b["act"]), dt, current_stream())); gemm(b["act"], w["wdown"], b["d"], wp=p["wdown"])
        gemm(b["h"], runner.w["lm_head"], b["logits"], True, wp=runner.wp["lm_head"] if runner.wp else None)

    def attn_only():
        st = current_stream()
        for li in range(s.layers):
            check(L.samd_tree_attention(_ptr(b["q"]), _ptr(runner.kv[li, 0]), _ptr(runner.kv[li, 1]), _ptr(b["attn"]), dt, R, s.heads,
                                        s.kv_heads, s.head_dim, runner.max_len, _ptr(runner.pf_mask), _ptr(v["cache_length"]), _ptr(runner.pf_n),
                                        runner.scale, _ptr(b["ws"]), b["ws_bytes"], st))

    def small_only():
        st = current_stream()
        sp_qkv = L.samd_gemm_splits((s.heads + 2 * s.kv_heads) * s.head_dim, s.hidden, RP) if native else 0
        sp_o = L.samd_gemm_splits(s.hidden, s.hidden, RP) if native else 0
        sp_gu = L.samd_gemm_splits(2 * s.inter, s.hidden, RP) if native else 0
        sp_d = L.samd_gemm_splits(s.hidden, s.inter, RP) if native else 0
        f = lambda sp, buf: (part if sp > 1 else buf)
        for li, w in enumerate(runner.w["layers"]):
            check(L.samd_rmsnorm(_ptr(b["x"]), _ptr(f(sp_d, b["d"])), _ptr(w["ln1"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, sp_d if sp_d > 1 else 0, RP * s.hidden, st))
            check(L.samd_rope_kv_write(_ptr(f(sp_qkv, b["qkv"])), _ptr(runner.pf_relpos), _ptr(v["cache_length"]), _ptr(runner.pf_n), _ptr(runner.cos), _ptr(runner.sin),
                                       _ptr(b["q"]), _ptr(runner.kv[li, 0]), _ptr(runner.kv[li, 1]), R, s.heads, s.kv_heads, s.head_dim, runner.max_len,
                                       runner.rope_rows, dt, sp_qkv if sp_qkv > 1 else 0, RP * (s.heads + 2 * s.kv_heads) * s.head_dim, st))
            check(L.samd_rmsnorm(_ptr(b["x"]), _ptr(f(sp_o, b["o"])), _ptr(w["ln2"]), _ptr(b["h"]), R, s.hidden, s.eps, dt, sp_o if sp_o > 1 else 0, RP * s.hidden, st))
            check(L.samd_silu_mul(_ptr(f(sp_gu, b["gu"])), _ptr(b["act"]), R, s.inter, dt, sp_gu if sp_gu > 1 else 0, RP * 2 * s.inter, st))

    res = {}
    parts = (("full", full),) if "--full-only" in sys.argv else (("full", full), ("gemms", gemms_only), ("attention+combine", attn_only), ("norm/rope/silu", small_only))
    for name, fn in parts:
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        res[name] = hip_time_ms(g.replay, 10)
    print(f"R={R} native_gemm={native}: " + "  ".join(f"{k} {v_:.3f} ms" for k, v_ in res.items()),
          f" | sum of parts {sum(v_ for k, v_ in res.items() if k != 'full'):.3f} ms" + (f" | weights {runner.weight_bytes()/res['gemms']/1e6:.0f} GB/s in gemms" if "gemms" in res else ""))
