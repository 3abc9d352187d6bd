"""layer_probe.py -- the verify forward of one row bucket (default 8 rows, L = 800, Vicuna-7B shapes, random weights) replayed as a
hipGraph: what scripts/layer_counters.sh profiles (rocprofv3 --kernel-trace for durations and the gaps between launches, PMC passes
for the per-kernel wave counters).  usage: python scripts/layer_probe.py [rows] [L] [replays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip.llama import LlamaRunner
from bench import VICUNA_7B, hip_time_ms

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L0 = int(sys.argv[2]) if len(sys.argv) > 2 else 800
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
runner = LlamaRunner.random_init(dict(VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(4096)
runner.prefill(sess, torch.randint(3, 32000, (1, L0), device="cuda"))
torch.cuda.synchronize()
v = sess.device_views()
runner.pf_n.fill_(max(1, R - 1))


def full():
    runner.forward_rows(R, runner.pf_tokens, runner.pf_relpos, runner.pf_mask, v["cache_length"], runner.pf_n)


full(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    full()
ms = hip_time_ms(g.replay, reps)
print(f"rows {R} L {L0}: forward {ms:.4f} ms = {ms * 1e3 / runner.shape.layers:.2f} us per layer incl. embed / lm_head", flush=True)
