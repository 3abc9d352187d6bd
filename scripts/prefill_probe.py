"""prefill_probe.py -- wall time of the one-pass (wide) prefill of a 512-token prompt at Vicuna-7B shapes, to compare with the sum of
its kernels' durations under rocprofv3 --stats (is the prefill host-bound?).  usage: python3 scripts/prefill_probe.py [tokens]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch, samd_hip, bench
from samd_hip.llama import LlamaRunner
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
runner = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(4096)
ids = torch.randint(3, 32000, (1, N), device="cuda")
runner.prefill(sess, ids); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    runner.prefill(sess, ids)
torch.cuda.synchronize()
print(f"prefill of {N} tokens: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms wall per call (6 calls in this process)")
