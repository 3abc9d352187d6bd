"""norm_fold_bench.py -- one decoder layer's projection + norm launches at 16 rows (Vicuna-7B shapes, cold weights: three sets in rotation),
the shipped sequence against the norm-fold sequence, as hipGraph replays timed with HIP events:
  shipped:   rmsnorm(partials) -> qkv_rope | o (split-K) -> rmsnorm(partials) -> pairs_silu | down (split-K)
  norm-fold: qkv_rope_norm | o complete sums + residual -> pairs_silu_norm | down complete sums + residual"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr as P, check
from bench import hip_time_ms
L = samd_hip.lib()
S = samd_hip.current_stream
H, D, hid, inter, max_len, R, eps = 32, 128, 4096, 11008, 2048, 16, 1e-6
sets = []
for _ in range(3):
    ws = {}
    w = (torch.randn((3 * hid, hid), device="cuda") * 0.02).half(); p = torch.empty_like(w)
    check(L.samd_gemm_pack_qkv64(P(w), P(p), 3 * H, hid, S())); ws["qkv"] = p
    for name, N, K in (("o", hid, hid), ("down", hid, inter)):
        w = (torch.randn((N, K), device="cuda") * 0.02).half(); p = torch.empty_like(w); pg = torch.empty_like(w)
        check(L.samd_gemm_pack_weights(P(w), P(p), N, K, S())); ws[name] = p
        check(L.samd_gemm_pack_groups(P(w), P(pg), N, K, S())); ws[name + "_g"] = pg
    w = (torch.randn((2 * inter, hid), device="cuda") * 0.02).half(); p = torch.empty_like(w)
    check(L.samd_gemm_pack_groups(P(w), P(p), 2 * inter, hid, S())); ws["gate_up"] = p
    del w
    sets.append(ws)
x = torch.randn((R, hid), device="cuda").half(); h = torch.zeros_like(x); attn = torch.randn((R, hid), device="cuda").half()
g1 = torch.ones(hid, device="cuda").half()
part = torch.zeros(8 * R * hid, device="cuda", dtype=torch.float32)
act = torch.zeros((R, inter), device="cuda", dtype=torch.float16)
q = torch.zeros((R, H, D), device="cuda", dtype=torch.float16)
kv = torch.zeros((2, H, max_len, D), device="cuda", dtype=torch.float16)
cs = torch.rand((64, D), device="cuda")
ssq = torch.ones((hid // 16, 16), device="cuda", dtype=torch.float32)
d_L = torch.tensor([800], dtype=torch.int32, device="cuda"); d_n = torch.tensor([13], dtype=torch.int32, device="cuda")
sp_o, sp_d = L.samd_gemm_splits(hid, hid, R), L.samd_gemm_splits(hid, inter, R)
F = samd_hip.F16

def shipped(w):
    check(L.samd_rmsnorm(P(x), P(part), P(g1), P(h), R, hid, eps, F, sp_d, R * hid, S()))
    check(L.samd_gemm_qkv_rope(P(h), P(w["qkv"]), R, hid, P(cs), P(d_L), P(d_n), P(q), P(kv[0]), P(kv[1]), H, H, D, max_len, F, S()))
    check(L.samd_gemm_skinny(P(attn), P(w["o"]), R, hid, hid, sp_o, P(part), None, F, S()))
    check(L.samd_rmsnorm(P(x), P(part), P(g1), P(h), R, hid, eps, F, sp_o, R * hid, S()))
    check(L.samd_gemm_pairs_silu(P(h), P(w["gate_up"]), R, inter, hid, P(act), F, S()))
    check(L.samd_gemm_skinny(P(act), P(w["down"]), R, hid, inter, sp_d, P(part), None, F, S()))

def folded(w):
    check(L.samd_gemm_qkv_rope_norm(P(x), P(ssq), P(g1), eps, P(w["qkv"]), R, hid, P(cs), P(d_L), P(d_n), P(q), P(kv[0]), P(kv[1]), H, H, D, max_len, F, S()))
    check(L.samd_gemm_cs_residual(P(attn), P(w["o_g"]), R, hid, hid, P(x), P(ssq), F, S()))
    check(L.samd_gemm_pairs_silu_norm(P(x), P(ssq), P(g1), eps, P(w["gate_up"]), R, inter, hid, P(act), F, S()))
    check(L.samd_gemm_cs_residual(P(act), P(w["down_g"]), R, hid, inter, P(x), P(ssq), F, S()))

singles = {
    "rmsnorm (8 partials)": lambda w: check(L.samd_rmsnorm(P(x), P(part), P(g1), P(h), R, hid, eps, F, sp_d, R * hid, S())),
    "qkv_rope": lambda w: check(L.samd_gemm_qkv_rope(P(h), P(w["qkv"]), R, hid, P(cs), P(d_L), P(d_n), P(q), P(kv[0]), P(kv[1]), H, H, D, max_len, F, S())),
    "qkv_rope_norm": lambda w: check(L.samd_gemm_qkv_rope_norm(P(x), P(ssq), P(g1), eps, P(w["qkv"]), R, hid, P(cs), P(d_L), P(d_n), P(q), P(kv[0]), P(kv[1]), H, H, D, max_len, F, S())),
    "o split-K": lambda w: check(L.samd_gemm_skinny(P(attn), P(w["o"]), R, hid, hid, sp_o, P(part), None, F, S())),
    "o complete + residual": lambda w: check(L.samd_gemm_cs_residual(P(attn), P(w["o_g"]), R, hid, hid, P(x), P(ssq), F, S())),
    "o complete + residual, 8 rows": lambda w: check(L.samd_gemm_cs_residual(P(attn), P(w["o_g"]), 8, hid, hid, P(x), P(ssq), F, S())),
    "down complete + residual, 8 rows": lambda w: check(L.samd_gemm_cs_residual(P(act), P(w["down_g"]), 8, hid, inter, P(x), P(ssq), F, S())),
    "pairs_silu": lambda w: check(L.samd_gemm_pairs_silu(P(h), P(w["gate_up"]), R, inter, hid, P(act), F, S())),
    "pairs_silu_norm": lambda w: check(L.samd_gemm_pairs_silu_norm(P(x), P(ssq), P(g1), eps, P(w["gate_up"]), R, inter, hid, P(act), F, S())),
    "down split-K": lambda w: check(L.samd_gemm_skinny(P(act), P(w["down"]), R, hid, inter, sp_d, P(part), None, F, S())),
    "down complete + residual": lambda w: check(L.samd_gemm_cs_residual(P(act), P(w["down_g"]), R, hid, inter, P(x), P(ssq), F, S())),
}

def timed(fn, reps=24):
    for w in sets:
        fn(w)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(reps):
            fn(sets[i % 3])
    return hip_time_ms(gr.replay, 10) / reps * 1e3

for name, fn in singles.items():
    print(f"{name:28s} {timed(fn):7.2f} us")
a, b = timed(shipped), timed(folded)
print(f"layer (projections + norms): shipped {a:.2f} us, norm-fold {b:.2f} us  ({a - b:+.2f} us per layer)")
