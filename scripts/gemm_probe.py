"""gemm_probe.py -- only the weight-streaming projections of ONE decoder layer as the runner launches them (Vicuna-7B shapes), rotating
over three weight sets so that nothing is served from the Infinity Cache; for rocprofv3 --stats / --pmc passes (scripts/pmc_gemm.sh).
At 16 rows the norm-fold launches: k_gemm_qkv_rope<NORM> (input norm + q|k|v + RoPE + K/V write), k_gemm_cs_residual (o, down: complete
sums + residual + sums of squares), k_gemm_pairs_silu<NORM> (post-attention norm + gate|up + SiLU); above, or with SAMD_NORM_FOLD=0:
k_gemm_qkv_rope, k_gemm_skinny (o, down; split-K 8), k_gemm_pairs_silu.
usage: python3 scripts/gemm_probe.py [layers_worth_of_launches] [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check

L = samd_hip.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, D, hid, inter, max_len = 32, 128, 4096, 11008, 2048
st = samd_hip.current_stream()
sets = []
for _ in range(3):
    ws = {}
    w = (torch.randn((3 * hid, hid), device="cuda") * 0.02).half(); p = torch.empty_like(w)
    check(L.samd_gemm_pack_qkv64(_ptr(w), _ptr(p), 3 * H, hid, st)); ws["qkv"] = p
    for name, N, K in (("o", hid, hid), ("down", hid, inter)):
        w = (torch.randn((N, K), device="cuda") * 0.02).half(); p = torch.empty_like(w); pg = torch.empty_like(w)
        check(L.samd_gemm_pack_weights(_ptr(w), _ptr(p), N, K, st)); ws[name] = p
        check(L.samd_gemm_pack_groups(_ptr(w), _ptr(pg), N, K, st)); ws[name + "_g"] = pg
    w = (torch.randn((2 * inter, hid), device="cuda") * 0.02).half(); p = torch.empty_like(w)
    check(L.samd_gemm_pack_groups(_ptr(w), _ptr(p), 2 * inter, hid, st)); ws["gate_up"] = p
    del w
    sets.append(ws)
A = {K: torch.randn((R, K), device="cuda").half() for K in (hid, inter)}
part = torch.zeros(8 * R * hid, device="cuda", dtype=torch.float32)
act = torch.zeros((R, inter), device="cuda", dtype=torch.float16)
q = torch.zeros((R, H, D), device="cuda", dtype=torch.float16)
kv = torch.zeros((2, H, max_len, D), device="cuda", dtype=torch.float16)
cs = torch.rand((64, D), device="cuda")
d_L = torch.tensor([800], dtype=torch.int32, device="cuda"); d_n = torch.tensor([max(1, R - 3)], dtype=torch.int32, device="cuda")
fold = R == 16 and os.environ.get("SAMD_NORM_FOLD", "1") != "0"
x = torch.randn((16, hid), device="cuda").half(); g1 = torch.ones(hid, device="cuda").half()
ssq = x.float().view(16, hid // 16, 16).pow(2).sum(-1).t().contiguous()
torch.cuda.synchronize()
nbytes = 0
for i in range(reps):
    w = sets[i % 3]
    if fold:
        check(L.samd_gemm_qkv_rope_norm(_ptr(x), _ptr(ssq), _ptr(g1), 1e-6, _ptr(w["qkv"]), 16, hid, _ptr(cs), _ptr(d_L), _ptr(d_n), _ptr(q), _ptr(kv[0]), _ptr(kv[1]), H, H, D, max_len, samd_hip.F16, st))
        check(L.samd_gemm_cs_residual(_ptr(A[hid]), _ptr(w["o_g"]), 16, hid, hid, _ptr(x), _ptr(ssq), samd_hip.F16, st))
        check(L.samd_gemm_pairs_silu_norm(_ptr(x), _ptr(ssq), _ptr(g1), 1e-6, _ptr(w["gate_up"]), 16, inter, hid, _ptr(act), samd_hip.F16, st))
        check(L.samd_gemm_cs_residual(_ptr(A[inter]), _ptr(w["down_g"]), 16, hid, inter, _ptr(x), _ptr(ssq), samd_hip.F16, st))
        x.copy_(A[hid])                                     # the residual stream would otherwise grow without bound over the repetitions
        nbytes += (3 * hid * hid + hid * hid + 2 * inter * hid + hid * inter) * 2
        continue
    check(L.samd_gemm_qkv_rope(_ptr(A[hid]), _ptr(w["qkv"]), R, hid, _ptr(cs), _ptr(d_L), _ptr(d_n), _ptr(q), _ptr(kv[0]), _ptr(kv[1]), H, H, D, max_len, samd_hip.F16, st))
    check(L.samd_gemm_skinny(_ptr(A[hid]), _ptr(w["o"]), R, hid, hid, L.samd_gemm_splits(hid, hid, R), _ptr(part), None, samd_hip.F16, st))
    check(L.samd_gemm_pairs_silu(_ptr(A[hid]), _ptr(w["gate_up"]), R, inter, hid, _ptr(act), samd_hip.F16, st))
    check(L.samd_gemm_skinny(_ptr(A[inter]), _ptr(w["down"]), R, hid, inter, L.samd_gemm_splits(hid, inter, R), _ptr(part), None, samd_hip.F16, st))
    nbytes += (3 * hid * hid + hid * hid + 2 * inter * hid + hid * inter) * 2
torch.cuda.synchronize()
print({"launches": reps * 4, "weight_bytes_per_layer": nbytes // reps})
