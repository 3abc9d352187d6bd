"""bench.py's synthetic request sources, run through the reference's draft rule on the CPU oracle (no GPU): what each source makes
samd_sam_only (max_predicts 60, alpha 4, len_bias 0, K 8) accept per step, and which row buckets its drafts fall into.  This pins the
calibration statements in bench.py's docstrings -- in particular the SUMMARIZATION-shaped source: profile
"readme_mat" must accept about what README.md:53 implies for that category (2.30 x 2.43 / 1.84 = 3.0), profile "copy_heavy" (mostly
prompt copies, span mean 16) must be carried by the dynamic automaton's sequence drafts and reach the 32/48/64-row buckets."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import sam_oracle as O  # noqa: E402

BUCKETS = (1, 8, 16, 32, 48, 64)


@pytest.fixture(scope="module")
def corpus():
    flat, off, docs = bench.synth_corpus(1 << 18)
    st = O.StaticSAM()
    fa, fp = O._i32(flat)
    oa, op = O._i64(off)
    O.lib().osam_add_batch(st._h, fp, op, len(off) - 1, bench.EOS)
    st.init_topk_next()
    return docs, st


def run_requests(st, reqs):
    """the scripted greedy verdict of bench.cpu_baseline: longest root->node path that follows the continuation"""
    dm = O.DraftModel(60, 4.0, 8, 0, sam_static=st)
    hist = {b: 0 for b in BUCKETS}
    by = {0: [0, 0], 1: [0, 0]}
    for prompt, target, new in reqs:
        dm.reset()
        dm.update(prompt)
        pos, done = len(prompt), 0
        while done < new:
            ty, tok, anc = dm.lookup_raw(target[pos])
            n = len(tok)
            hist[next(b for b in BUCKETS if n <= b)] += 1
            depth, ok, best = [0] * n, [True] * n, 0
            for i in range(1, n):
                depth[i] = depth[anc[i]] + 1
                ok[i] = ok[anc[i]] and tok[i] == target[pos + depth[i]]
                if ok[i] and depth[i] > depth[best]:
                    best = i
            a = depth[best] + 1
            dm.update(target[pos:pos + a])
            pos, done = pos + a, done + a
            by[ty][0] += 1
            by[ty][1] += a
    steps = by[0][0] + by[1][0]
    return (by[0][1] + by[1][1]) / steps, hist, by


def test_summarization_requests_have_the_stated_shape(corpus):
    docs, _ = corpus
    rng = np.random.default_rng(5)
    for name, params in bench.SUMM_PROFILES.items():
        for _ in range(5):
            prompt, target, new = bench.synth_request_summarization(rng, docs, **params)
            assert bench.SUMM_PROMPT_RANGE[0] <= len(prompt) <= bench.SUMM_PROMPT_RANGE[1]
            assert bench.SUMM_NEW_RANGE[0] <= new <= bench.SUMM_NEW_RANGE[1]
            assert target[:len(prompt)] == prompt and len(target) == len(prompt) + new + 64
            assert len(target) + 60 < 2048 + 64 and bench.EOS not in target and min(target) >= 3 and max(target) < bench.VOCAB
            # fits Vicuna's max_cache_len with the reference's loop guard (SO/samd_model.py:217)
            assert len(prompt) + new + 60 < 2048
    a = bench.synth_request_summarization(np.random.default_rng(9), docs)
    b = bench.synth_request_summarization(np.random.default_rng(9), docs)
    assert a == b                                                              # seeded


def test_summarization_profiles_accept_what_their_docstrings_say(corpus):
    docs, st = corpus
    out = {}
    for name, params in bench.SUMM_PROFILES.items():
        rng = np.random.default_rng(2000)
        reqs = [bench.synth_request_summarization(rng, docs, **params) for _ in range(8)]
        out[name] = run_requests(st, reqs)
    mat, hist, by = out["readme_mat"]
    assert 2.6 <= mat <= 3.7, mat                                              # README-implied ~3.0 (2.30 x 2.43 / 1.84)
    assert by[0][1] > by[1][1]                                                 # ... and still most TOKENS come from dyn sequence drafts
    mat_c, hist_c, by_c = out["copy_heavy"]
    assert 4.5 <= mat_c <= 7.5, mat_c
    assert by_c[0][1] > 8 * by_c[1][1]                                         # dyn-SAM sequence drafts dominate, as on CNN/DM
    wide = hist_c[32] + hist_c[48] + hist_c[64]
    assert wide >= 0.2 * sum(hist_c.values()), hist_c                          # the 32/48/64-row buckets are a real share of the steps
    assert by_c[0][1] / by_c[0][0] > 6.0                                       # a sequence step accepts a long prefix


def test_headline_source_is_calibrated_to_the_published_mean_accepted_tokens(corpus):
    """bench.synth_request: ~2.3 accepted tokens per step (README.md:53), 8/16-row buckets dominate"""
    docs, st = corpus
    rng = np.random.default_rng(1000)
    reqs = []
    for _ in range(4):
        prompt, target = bench.synth_request(rng, docs)
        reqs.append((prompt, target, 512))
    mat, hist, _ = run_requests(st, reqs)
    assert 2.0 <= mat <= 2.8, mat
    assert hist[8] + hist[16] >= 0.85 * sum(hist.values())
