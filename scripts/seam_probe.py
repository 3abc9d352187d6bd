"""seam_probe.py -- VERDICT r03 #2's experiment, measured: o_proj launched on a second queue BESIDE attention (it requests its weights at
entry and polls a device counter the merge launch arrives on before it reads its activations) against the product's serial launches.
One "layer" = k_tree_attention + k_attn_combine + k_gemm_cs_residual<8> (o_proj), 32 layers with their own K/V and weights,
Vicuna-7B geometry, 8-row bucket (7 draft nodes), L = 800.
   serial     one stream, one hipGraph: attention, merge, o_proj per layer                      (what the runner does)
   two-queue  graph A (attention + signalling merge per layer) on stream A, graph B (early o_proj per layer) on stream B, no edges
   A alone    graph A by itself                                                                  (what the overlap costs attention)
usage: python scripts/seam_probe.py [L]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip import _ptr, check, lib

L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 800
H, D, layers, max_len, R, hidden = 32, 128, 32, 2048, 8, 4096
Lib = lib()
dev = "cuda"
kv = torch.randn((layers, 2, H, max_len, D), device=dev).half()
q = torch.randn((16, H, D), device=dev).half()
attn = torch.zeros((layers, 16, H, D), device=dev, dtype=torch.float16)      # one per layer: graph A may run ahead of graph B
x = torch.zeros((16, hidden), device=dev, dtype=torch.float16)
ssq = torch.zeros((hidden // 16, 16), device=dev, dtype=torch.float32)
mask = torch.tensor([(1 << (i + 1)) - 1 if i < 63 else -1 for i in range(64)], dtype=torch.int64, device=dev)
d_L = torch.tensor([L0], dtype=torch.int32, device=dev); d_n = torch.tensor([R - 1], dtype=torch.int32, device=dev)
ws = torch.zeros(Lib.samd_tree_attention_workspace(R, H, D), dtype=torch.uint8, device=dev)
wo = [torch.empty((hidden, hidden), device=dev, dtype=torch.float16) for _ in range(layers)]
raw = (torch.randn((hidden, hidden), device=dev) * 0.02).half()
for w in wo:
    check(Lib.samd_gemm_pack_groups(_ptr(raw), _ptr(w), hidden, hidden, samd_hip.current_stream()))
counter = torch.zeros(layers, dtype=torch.int32, device=dev)
epoch = torch.zeros((layers, hidden // 16), dtype=torch.int32, device=dev)
scale = 1.0 / math.sqrt(D)
arrivals = R * H
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def st(s):
    return samd_hip.C.c_void_p(s.cuda_stream)


def serial(s):
    for li in range(layers):
        check(Lib.samd_tree_attention(_ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(attn[li]), samd_hip.F16, R, H, H, D, max_len, _ptr(mask), _ptr(d_L), _ptr(d_n),
                                      scale, _ptr(ws), ws.numel(), st(s)))
        check(Lib.samd_gemm_cs_residual(_ptr(attn[li]), _ptr(wo[li]), 8, hidden, hidden, _ptr(x), _ptr(ssq), samd_hip.F16, st(s)))


def branch_a(s):
    for li in range(layers):
        check(Lib.samd_tree_attention_signal(_ptr(q), _ptr(kv[li, 0]), _ptr(kv[li, 1]), _ptr(attn[li]), samd_hip.F16, R, H, H, D, max_len, _ptr(mask), _ptr(d_L),
                                             _ptr(d_n), scale, _ptr(ws), ws.numel(), _ptr(counter[li:]), st(s)))


def branch_b(s):
    for li in range(layers):
        check(Lib.samd_gemm_cs_residual_early(_ptr(attn[li]), _ptr(wo[li]), hidden, hidden, _ptr(x), _ptr(ssq), samd_hip.F16, _ptr(counter[li:]),
                                              _ptr(epoch[li]), arrivals, st(s)))


def capture(fn, s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(s); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn(s)
    return g


# reference result of one layer's o_proj for the correctness check of the early form
torch.cuda.synchronize()
g_serial = capture(serial, sA)
x.zero_(); g_serial.replay(); torch.cuda.synchronize(); want = x.clone()
counter.zero_(); epoch.zero_()
# warm both branches eagerly once (branch A first so that B's polls are satisfied), then capture each on its own stream
with torch.cuda.stream(sA):
    branch_a(sA)
with torch.cuda.stream(sB):
    branch_b(sB)
torch.cuda.synchronize()
gA = torch.cuda.CUDAGraph(); gB = torch.cuda.CUDAGraph()
with torch.cuda.stream(sA):
    with torch.cuda.graph(gA, stream=sA):
        branch_a(sA)
with torch.cuda.stream(sB):
    with torch.cuda.graph(gB, stream=sB):
        branch_b(sB)
torch.cuda.synchronize()


def timed(run, reps=20):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6 / layers


def two_queue():
    with torch.cuda.stream(sB):
        gB.replay()
    with torch.cuda.stream(sA):
        gA.replay()


def a_alone():
    with torch.cuda.stream(sA):
        gA.replay()


def ser():
    with torch.cuda.stream(sA):
        g_serial.replay()


x.zero_(); two_queue(); torch.cuda.synchronize()
ok = torch.equal(x, want)
t_ser, t_two, t_a = timed(ser), timed(two_queue), None
# graph A alone leaves the counters ahead of B's epochs; re-align afterwards
t_a = timed(a_alone)
print(f"L = {L0}, {R} rows: serial {t_ser:.2f} us per layer | two queues {t_two:.2f} us per layer | attention + merge alone {t_a:.2f} us per layer | "
      f"early o_proj result {'identical' if ok else 'DIFFERENT'}", flush=True)
