"""sdpa_probe.py -- PyTorch's fused causal attention on the prefill's shape (32 heads x N x 128, fp16) per backend"""
import sys, torch
from torch.nn.attention import sdpa_kernel, SDPBackend
F = torch.nn.functional
def t_ms(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for N in [int(x) for x in sys.argv[1:]] or [512, 1024, 1280, 1536]:
    q = torch.randn(1, 32, N, 128, device="cuda", dtype=torch.float16)
    # the runner's layout: [N, H, D] storage viewed as [1, H, N, D]
    qt = torch.randn(N, 32, 128, device="cuda", dtype=torch.float16).transpose(0, 1)[None]
    out = []
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            with sdpa_kernel([be]):
                a = t_ms(lambda: F.scaled_dot_product_attention(q, q, q, is_causal=True))
                b = t_ms(lambda: F.scaled_dot_product_attention(qt, q, q, is_causal=True))
            out.append(f"{name} {a * 1e3:.0f} us (q as [N,H,D] view: {b * 1e3:.0f})")
        except Exception as e:
            out.append(f"{name}: {type(e).__name__} {str(e)[:60]}")
    d = t_ms(lambda: F.scaled_dot_product_attention(q, q, q, is_causal=True))
    print(f"N={N}: default {d * 1e3:.0f} us; " + "; ".join(out), flush=True)
