"""gemm_shapes_probe.py -- what the LIBRARY GEMM (torch.mm -> hipBLASLt) reaches on the four projection shapes of a Vicuna-7B decoder
layer at prefill row counts, and the other launches of the wide prefill (SDPA, our glue kernels): PFLOP/s per shape, ms per layer.
usage: python3 scripts/gemm_shapes_probe.py [rows ...]
SAMD_PROBE_TUNABLE=<csv>: PyTorch's TunableOp enabled and tuning (every hipBLASLt / rocBLAS solution timed per shape), results written to <csv>."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
if os.environ.get("SAMD_PROBE_TUNABLE"):
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True); tun.set_filename(os.environ["SAMD_PROBE_TUNABLE"])
    if not os.environ.get("SAMD_PROBE_TUNABLE_ENV"):                  # (else: duration / iterations / rotating buffer from the PYTORCH_TUNABLEOP_* variables)
        tun.set_max_tuning_duration(20); tun.set_max_tuning_iterations(20)
rows = [int(x) for x in sys.argv[1:]] or [512, 1024, 1280, 1536, 2048]
H, I = 4096, 11008
shapes = {"qkv": (3 * H, H), "o": (H, H), "gate|up": (2 * I, H), "down": (H, I)}
def t_ms(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for M in rows:
    tot_ms, tot_fl = 0.0, 0.0
    line = []
    for name, (N, K) in shapes.items():
        a = torch.randn(M, K, device="cuda", dtype=torch.float16) * 0.1
        ws = [torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02 for _ in range(3)]     # rotate: nothing served from MALL
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        k = [0]
        def f():
            torch.mm(a, ws[k[0] % 3].t(), out=out); k[0] += 1
        ms = t_ms(f)
        fl = 2.0 * M * N * K
        tot_ms += ms; tot_fl += fl
        line.append(f"{name} {ms * 1e3:.0f} us {fl / ms / 1e12:.2f} PF/s")
    q = torch.randn(1, 32, M, 128, device="cuda", dtype=torch.float16)
    att = t_ms(lambda: torch.nn.functional.scaled_dot_product_attention(q, q, q, is_causal=True))
    print(f"M={M}: " + "; ".join(line) + f"; GEMMs {tot_ms:.3f} ms/layer = {tot_fl / tot_ms / 1e12:.2f} PF/s ({tot_ms * 32:.1f} ms per 32 layers); SDPA causal {att * 1e3:.0f} us/layer")
