"""where the per-step host time goes: graph replays back to back vs replay+sync vs the full generate() loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import numpy as np, torch
import samd_hip, samd_sam_only as SO
from samd_hip.engine import ScriptedAcceptance, StepReport
from samd_hip.llama import LlamaRunner
import bench
flat, off, docs = bench.synth_corpus(1 << 20)
auto = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0).upload()
runner = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
lm = ScriptedAcceptance(runner, bench.VOCAB, 2048)
cfg = SO.SamdConfig(max_predicts=60, alpha=4.0, K=8, len_bias=0)
model = SO.SamdModel(cfg, lm, SO.DraftModel(cfg, sam_static=SO.sam.StaticSAM._from_automaton(auto), device="cuda"), 2, torch.float16, "cuda")
gcfg = SO.SamdGenerationConfig(max_new_tokens=512, max_cache_len=2048)
rng = np.random.default_rng(5)
prompt, target = bench.synth_request(rng, docs)
lm.set_target(target)
ids = torch.tensor([prompt], device="cuda")
# (c) full loop
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for new_ids, r in model._run(ids, gcfg, 150):
        n += 1
    torch.cuda.synchronize(); tc = (time.perf_counter() - t0)
print(f"full loop incl. start(): {tc / n * 1e3:.3f} ms/step over {n} steps")
eng = model.engine
rep0 = eng.start(ids)
R = 16
g = eng._graphs.get(R) or eng._capture(R)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 50
eng.start(ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    g.replay(); torch.cuda.current_stream().synchronize(); StepReport(eng._report_np)
tb = (time.perf_counter() - t0) / 50
print(f"R=16: back-to-back replays {ta*1e3:.3f} ms/step; replay+sync+report {tb*1e3:.3f} ms/step")
