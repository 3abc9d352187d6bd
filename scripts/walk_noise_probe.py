"""walk_noise_probe.py -- the traversal kernel on the Zipfian / headline corpus at several noise rates (0 = pure corpus copies: register path, chain
words and real branching only; 0.1 = the bench's): what the climbs of the mismatching tokens cost a lock-step wave.  usage: python scripts/walk_noise_probe.py [dist] [log2 tokens]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch, bench, samd_hip
dist = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n_tok = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
flat, off, docs = bench.synth_corpus(n_tok) if dist == "markov" else bench.synth_corpus_zipf(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, samd_hip.KIND_COUNT).upload()
cdf = None if dist == "markov" else bench.zipf_cdf(bench.VOCAB)
for p in (0.0, 0.02, 0.05, 0.10, 0.20):
    roof, _ = bench.walk_roofline(sam, docs, np.random.default_rng(7), 1 << 20, 16, 10, n_tok, slots_per_pair=16, noise_cdf=cdf, noise_p=p)
    print(f"{dist} 2^{int(np.log2(n_tok))} noise {p:.2f}: {roof['launch_ms']:.4f} ms, visited/token {roof['visited_states'] / (16 << 20):.3f}, frac {roof['frac']:.4f}, "
          f"{roof['transitions_per_s'] / 1e9:.1f} G transitions/s", flush=True)
