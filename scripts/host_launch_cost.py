"""host_launch_cost.py -- host time of one ctypes kernel launch through the binding, piece by piece (no GPU wait inside the loop):
current_stream(), _ptr(), the library call itself, a torch in-place op, a graph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch, samd_hip
from samd_hip import _ptr, check, current_stream
L = samd_hip.lib()
part = torch.zeros((2, 16, 4096), device="cuda"); bias = torch.zeros(4096, device="cuda", dtype=torch.float16); out = torch.zeros((16, 4096), device="cuda", dtype=torch.float16)
N = 2000
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    dt = (time.perf_counter() - t0) / N * 1e6; torch.cuda.synchronize(); return dt
print(f"current_stream(): {t(current_stream):.2f} us")
print(f"_ptr(tensor): {t(lambda: _ptr(part)):.2f} us")
st = current_stream(); pp, pb, po = _ptr(part), _ptr(bias), _ptr(out)
print(f"library launch, prepared args: {t(lambda: L.samd_sum_partials_bias(pp, 2, 16 * 4096, pb, po, 8, 4096, samd_hip.F16, st)):.2f} us")
print(f"library launch via check + _ptr + current_stream: {t(lambda: check(L.samd_sum_partials_bias(_ptr(part), 2, 16 * 4096, _ptr(bias), _ptr(out), 8, 4096, samd_hip.F16, current_stream()))):.2f} us")
x = torch.zeros(1, device="cuda", dtype=torch.int32)
print(f"torch x.add_(1): {t(lambda: x.add_(1)):.2f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(15): L.samd_sum_partials_bias(pp, 2, 16 * 4096, pb, po, 8, 4096, samd_hip.F16, current_stream())
print(f"graph replay (15 nodes): {t(g.replay):.2f} us")
