"""host_turnaround.py -- where the host spends the time between two decode steps (bench.py's default workload, scripted acceptance):
g.replay() (hipGraphLaunch), the stream synchronisation (= the step on the GPU + wake-up) and the Python between the wake-up and the next
replay (report parsing, truncation, bookkeeping, the generator hop).  The GPU idles for the Python part plus the launch latency."""
import os, sys, time
os.environ["SAMD_REPORT_PUSH"] = "0"      # this script times the copy + synchronise form of a step (it replaces DecodeEngine.step by its own);
                                          # the pushed report of round 4 is measured by bench.py A/B (SAMD_REPORT_PUSH=0|1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import numpy as np
import torch
import samd_hip
import bench
import samd_sam_only as SO
from samd_hip.engine import ScriptedAcceptance
from samd_hip.llama import LlamaRunner

samd_hip.host_waits_by_spinning()
flat, off, docs = bench.synth_corpus(1 << 20)
auto = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, samd_hip.KIND_COUNT).upload()
sam = SO.sam.StaticSAM._from_automaton(auto)
cfg = SO.SamdConfig(max_predicts=60, alpha=4.0, K=8, len_bias=0)
runner = LlamaRunner.random_init(dict(bench.VICUNA_7B), 2048, torch.float16, seed=0)
lm = ScriptedAcceptance(runner, bench.VOCAB, 2048)
model = SO.SamdModel(cfg, lm, SO.DraftModel(cfg, sam_static=sam, device="cuda"), bench.EOS, torch.float16, "cuda")
gcfg = SO.SamdGenerationConfig(max_new_tokens=512, max_cache_len=2048)
model.set_cache(gcfg)
eng = model.engine
T = {"replay": 0.0, "sync": 0.0, "n": 0}
orig_step = eng.step
def step(n_next):
    R = eng.verifier.bucket(n_next)
    eng.bucket_steps[R] = eng.bucket_steps.get(R, 0) + 1
    g = eng._graphs.get(R) or eng._capture(R)
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize(); t2 = time.perf_counter()
    T["replay"] += t1 - t0; T["sync"] += t2 - t1; T["n"] += 1
    from samd_hip.engine import StepReport
    return StepReport(eng._report_np)
eng.step = step
rng = np.random.default_rng(5)
total = 0.0; steps = 0
for r in range(4):
    prompt, target = bench.synth_request(rng, docs)
    lm.set_target(target)
    ids = torch.tensor([prompt], dtype=torch.long, device="cuda")
    it = model._run(ids, gcfg, 512)
    next(it)
    T.update(replay=0.0, sync=0.0, n=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 0
    for _ in it:
        k += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"request {r}: {k} steps, wall {dt / k * 1e6:.1f} us per step = replay call {T['replay'] / T['n'] * 1e6:.1f} + sync wait {T['sync'] / T['n'] * 1e6:.1f} + python {(dt - T['replay'] - T['sync']) / k * 1e6:.1f}")
