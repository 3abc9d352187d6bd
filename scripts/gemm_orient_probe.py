"""gemm_orient_probe.py -- does the library do better on the TRANSPOSED problem (out^T = W x^T) at prefill row counts?"""
import sys, torch
def t_us(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
H, I = 4096, 11008
shapes = {"qkv": (3 * H, H), "o": (H, H), "gate|up": (2 * I, H), "down": (H, I)}
for M in [int(x) for x in sys.argv[1:]] or [512, 1024]:
    line = []
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.float16) * 0.1
        ws = [torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02 for _ in range(3)]
        out, outT = torch.empty(M, N, device="cuda", dtype=torch.float16), torch.empty(N, M, device="cuda", dtype=torch.float16)
        k = [0]
        def a(): torch.mm(x, ws[k[0] % 3].t(), out=out); k[0] += 1
        def b(): torch.mm(ws[k[0] % 3], x.t(), out=outT); k[0] += 1
        xt = x.t().contiguous()
        def c(): torch.mm(ws[k[0] % 3], xt, out=outT); k[0] += 1
        # K split in two accumulating calls (addmm): two launches with half the k range each
        h = K // 2 // 64 * 64
        def d():
            w = ws[k[0] % 3]; k[0] += 1
            torch.mm(x[:, :h], w[:, :h].t(), out=out); out.addmm_(x[:, h:], w[:, h:].t())
        line.append(f"{name}: x W^T {a.__call__() or t_us(a):.0f} | W x^T {t_us(b):.0f} | W xT(contig) {t_us(c):.0f} | two k halves {t_us(d):.0f}")
    print(f"M={M}: " + "; ".join(line), flush=True)
