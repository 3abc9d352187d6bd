"""walk_probe.py -- only the SAM traversal kernel (k_static_walk) on the bench's corpus, for rocprofv3 PMC passes.
usage: python3 scripts/walk_probe.py [corpus_tokens] [streams] [tokens_per_stream] [launches] [dist: markov|zipf] [slots_per_pair]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch
import bench
import samd_hip

n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dist = sys.argv[5] if len(sys.argv) > 5 else "markov"
slots = int(sys.argv[6]) if len(sys.argv) > 6 else bench.WALK_BIGRAM_SLOTS_PER_PAIR
flat, off, docs = bench.synth_corpus(n_tok) if dist == "markov" else bench.synth_corpus_zipf(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, samd_hip.KIND_COUNT).upload()
roof, _ = bench.walk_roofline(sam, docs, np.random.default_rng(7), B, T, launches, n_tok, slots_per_pair=slots,
                              noise_cdf=None if dist == "markov" else bench.zipf_cdf(bench.VOCAB))
print(roof)
