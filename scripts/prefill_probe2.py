"""prefill_probe2.py -- the request start over prompt lengths, with and without round 5's two library-shaping measures (LlamaRunner._prefill_wide:
attention rows padded to a multiple of 128, projections split at a tile-quantisation step), and that both leave the prompt's logits where
they were (max |d logit| against the plain form).  usage: python3 scripts/prefill_probe2.py [lengths ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import samd_hip
from samd_hip.llama import LlamaRunner
from bench import VICUNA_7B
lengths = [int(x) for x in sys.argv[1:]] or [512, 1100, 1237, 1280, 1333, 1408, 1501, 1536]
runner = LlamaRunner.random_init(dict(VICUNA_7B), 2048, torch.float16, seed=0)
sess = samd_hip.Session(2048)
t0 = time.perf_counter(); runner.tune_prefill(2048); torch.cuda.synchronize()
print(f"tune_prefill: {1e3 * (time.perf_counter() - t0):.0f} ms; splits {runner.prefill_plan_summary()}")
plan = runner._pf_plan
g = torch.Generator().manual_seed(1)
def run(ids, reps=3):
    best = 1e9
    for _ in range(reps):
        sess.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        logits = runner.prefill(sess, ids); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3, logits.float().clone(), runner.kv[:, :, :, :ids.numel()].float().clone()
for n in lengths:
    ids = torch.randint(3, 32000, (1, n), generator=g).cuda()
    runner._pf_plan, LlamaRunner.PF_ATTN_PAD = {}, 1
    a_ms, a_log, a_kv = run(ids)
    runner._pf_plan, LlamaRunner.PF_ATTN_PAD = {}, 128
    b_ms, b_log, b_kv = run(ids)
    runner._pf_plan = plan
    c_ms, c_log, c_kv = run(ids)
    print(f"N={n}: plain {a_ms:.2f} ms | attention padded {b_ms:.2f} (|d logit| {float((a_log - b_log).abs().max()):.4f}, |d kv| {float((a_kv - b_kv).abs().max()):.4f}) | "
          f"+ row splits {c_ms:.2f} (|d logit| {float((a_log - c_log).abs().max()):.4f}, |d kv| {float((a_kv - c_kv).abs().max()):.4f}; "
          f"splits qkv {runner._pf_split('wqkv', n)} gate|up {runner._pf_split('wgu', n)} o {runner._pf_split('wo', n)} down {runner._pf_split('wdown', n)})", flush=True)
