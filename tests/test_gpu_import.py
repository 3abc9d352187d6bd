"""SURVEY section 8(f) rank 1 on the GPU: a pickle written by the reference's own dump_sam (samd_sam_only/sam/utils.py:20-39) and an
image written by the corpus -> automaton CLI (tools/gen_sam_alpaca_sam_only.py:15-49) are uploaded and WALKED: every lookup,
committed transition, batched walk and best-first tree draft must equal the oracle's on an automaton built from the same documents."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from oracle import sam_oracle as O
from test_gen_sam_cpu import corpus_documents, run_cli
from util import toy_dialogues

HERE = os.path.dirname(os.path.abspath(__file__))


def query_streams(rng, docs, vocab, n_streams=6, length=48):
    """copied document spans glued with a few random tokens (incl. one the corpus never has)"""
    out = []
    for _ in range(n_streams):
        s = []
        while len(s) < length:
            d = docs[int(rng.integers(0, len(docs)))]
            a = int(rng.integers(0, len(d)))
            s += d[a:a + int(rng.integers(2, 12))] + rng.integers(0, vocab + 2, int(rng.integers(0, 3))).tolist()
        out.append([int(t) for t in s[:length]])
    return out


def check_walks_and_drafts(sam, ora, docs, vocab, seed):
    """`sam`: samd_sam_only.StaticSAM on the GPU; `ora`: the oracle's automaton of the same documents"""
    rng = np.random.default_rng(seed)
    streams = query_streams(rng, docs, vocab)
    visited = []
    for s in streams:
        sam.reset(); ora.reset()
        for t in s:
            assert tuple(sam.lookup(t)) == tuple(ora.lookup(t))                 # static_sam.py:122-125, no side effect
            sam.transfer_tokens([t]); ora.transfer_tokens([t])                   # static_sam.py:118-120
            assert (sam.cur_index, sam.cur_length) == tuple(ora.cursor())
            visited.append((ora.cur_index, ora.cur_length))
    # the batched walk kernel (chain words) over the same streams, one lane each, traced
    auto = sam._automaton()
    B, T = len(streams), len(streams[0])
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    toks = torch.tensor(streams, dtype=torch.int32, device="cuda").t().contiguous()       # time-major
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    auto.walk(cur, toks, commit=True, trace=trace)
    trace = trace.cpu()
    for b, s in enumerate(streams):
        ora.reset()
        for t, tk in enumerate(s):
            ora.transfer_tokens([tk])
            assert tuple(trace[t, b].tolist()) == tuple(ora.cursor()), (b, t)
    # best-first tree drafts from visited states (static_sam.py:182-215), several sizes
    pick = [visited[int(i)] for i in rng.integers(0, len(visited), 24)] + [(0, 0)]
    for idx, match in pick:
        for mp, alpha in ((60, 4.0), (17, 1.0)):
            sam.max_predicts = ora.max_predicts = mp
            sam.alpha = ora.alpha = alpha
            got_tokens, got_buf = sam.gen_draft(idx, match, 5)
            want_tokens, want_buf = ora.gen_draft(idx, match, 5)
            assert got_tokens == want_tokens, (idx, match, mp)
            assert got_buf["tree_position_ids"].view(-1).tolist() == np.asarray(want_buf["tree_position_ids"]).reshape(-1).tolist()
            assert got_buf["tree_retrieve_indices"].tolist() == np.asarray(want_buf["tree_retrieve_indices"]).tolist()
            assert got_buf["tree_attn_mask"].view(len(got_tokens), -1).int().tolist() == np.asarray(want_buf["tree_attn_mask"]).reshape(len(got_tokens), -1).astype(int).tolist()


def test_reference_pickle_walks_on_the_gpu(golden):
    import samd_sam_only as SO
    meta = golden("ref_static_sam_docs.json.gz")
    sam = SO.load_sam(os.path.join(HERE, "golden", "ref_static_sam.pkl"))          # the reference's object graph -> flat image
    ora = O.StaticSAM.build(meta["docs"], meta["eos"])
    assert len(sam.states) == meta["n_states"] == ora.num_states
    check_walks_and_drafts(sam, ora, meta["docs"], 30, seed=3)


@pytest.mark.parametrize("with_data", [True, False])
def test_cli_image_walks_on_the_gpu(tmp_path, with_data):
    import samd_sam_only as SO
    rng = np.random.default_rng(11)
    tok_dir, data_path, sam_path = run_cli(str(tmp_path), lambda words: toy_dialogues(rng, words), data=with_data)
    docs, eos = corpus_documents(tok_dir, data_path if with_data else None)
    sam = SO.load_sam(sam_path)
    ora = O.StaticSAM.build(docs, eos)
    check_walks_and_drafts(sam, ora, docs, 123, seed=5)
    # and through DraftModel, as inference_sam_only.py would use it: lookups follow the oracle's DraftModel step by step
    cfg = SO.SamdConfig(max_predicts=40, alpha=4.0, K=8, len_bias=0)
    dm = SO.DraftModel(cfg, sam_static=sam, device="cuda")
    od = O.DraftModel(max_predicts=40, alpha=4.0, K=8, len_bias=0, sam_dyn=O.DynSAM(), sam_static=ora)
    dm.reset(); od.reset()
    stream = query_streams(np.random.default_rng(9), docs, 123, n_streams=1, length=40)[0]
    for i in range(0, len(stream) - 4, 4):
        chunk = stream[i:i + 4]
        dm.update(torch.tensor(chunk, device="cuda")); od.update(chunk)
        kind, toks, _ = dm.lookup(stream[i + 4])
        okind, otoks, _ = od.lookup(stream[i + 4])
        assert (kind.name, toks) == (okind, otoks)
