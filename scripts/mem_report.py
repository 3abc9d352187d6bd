import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/sam-decoding_amd']
import torch, bench
from samd_hip.llama import LlamaRunner
r = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
a = r.memory_report()
print("default", {k: round(v / 1e9, 2) for k, v in a.items()})
assert r.release_row_major()
b = r.memory_report()
print("memory-first", {k: round(v / 1e9, 2) for k, v in b.items()}, "max_draft_rows", r.max_draft_rows())
