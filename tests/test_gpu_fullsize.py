"""GPU checks at BASELINE.json's full sizes (configs[1] as bench.py runs it: 2^22-token corpus, ~4.3 M states, 373 MB
automaton; 2^18 cursors here instead of 2^20 to keep the test short -- the kernel and the data are the same).

Bit-exact parity against the C oracle on a sample, plus size-independent properties of a longest-suffix-match walk that
need no oracle: composition (16 tokens == 8 + 8 with committed cursors), substring streams match their full length, the
reported suffix occurs in the corpus and the one-longer suffix does not, and an image round trip."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the bench's own generators: same corpus, same request process)
import samd_hip  # noqa: E402
from oracle import sam_oracle as O  # noqa: E402

N_TOKENS = 1 << 22
B, T = 1 << 18, 16


@pytest.fixture(scope="module")
def corpus():
    flat, off, docs = bench.synth_corpus(N_TOKENS)
    sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, samd_hip.KIND_COUNT).upload()
    return flat, off, docs, sam


def _streams(docs, rng, noise):
    n_docs, doc_len = docs.shape
    d = rng.integers(0, n_docs, B)
    s = rng.integers(0, doc_len - T, B)
    toks = docs[d[None, :], (s[None, :] + np.arange(T)[:, None])]
    if noise:
        toks = np.where(rng.random((T, B)) < noise, rng.integers(3, bench.VOCAB, (T, B)), toks)
    return np.ascontiguousarray(toks.astype(np.int32))


def _walk(sam, toks, cursors=None, trace=False):
    d_toks = torch.from_numpy(toks).cuda()
    cur = torch.zeros((toks.shape[1], 2), dtype=torch.int32, device="cuda") if cursors is None else cursors
    tr = torch.zeros((toks.shape[0], toks.shape[1], 2), dtype=torch.int32, device="cuda") if trace else None
    sam.walk(cur, d_toks, commit=True, trace=tr)
    torch.cuda.synchronize()
    return cur, tr


def test_full_size_walk_matches_oracle_on_a_sample(corpus):
    flat, off, docs, sam = corpus
    assert sam.info()["n_states"] > 4_000_000
    toks = _streams(docs, np.random.default_rng(11), 0.10)
    cur, tr = _walk(sam, toks, trace=True)
    tr = tr.cpu().numpy()
    ora = O.StaticSAM.build([flat[off[i]:off[i + 1]] for i in range(len(off) - 1)], bench.EOS)
    assert ora.num_states == sam.info()["n_states"]
    for b in np.random.default_rng(12).integers(0, B, 1500):
        idx, ln = 0, 0
        for t in range(T):
            idx, ln = ora.transfer_state(idx, ln, int(toks[t, b]))
            assert (idx, ln) == tuple(tr[t, b]), (b, t)


def test_full_size_walk_composes_and_matches_substrings(corpus):
    flat, off, docs, sam = corpus
    rng = np.random.default_rng(21)
    noisy = _streams(docs, rng, 0.10)
    whole, _ = _walk(sam, noisy)
    first, _ = _walk(sam, noisy[:8])
    both, _ = _walk(sam, np.ascontiguousarray(noisy[8:]), cursors=first)
    assert torch.equal(whole, both)                                    # walk(a + b) == walk(b) from the cursor of walk(a)
    clean = _streams(docs, rng, 0.0)
    _, tr = _walk(sam, clean, trace=True)
    assert (tr[:, :, 1].cpu().numpy() == np.arange(1, T + 1)[:, None]).all()      # a substring of the corpus matches in full


def test_full_size_reported_suffix_is_the_longest_one_in_the_corpus(corpus):
    flat, off, docs, sam = corpus
    toks = _streams(docs, np.random.default_rng(31), 0.25)
    cur, _ = _walk(sam, toks)
    cur = cur.cpu().numpy()
    # the automaton is built over documents joined by EOS, so occurrences may not span a document end: search inside docs
    text = np.ascontiguousarray(docs.astype(np.int32))
    raw = text.tobytes()
    row_bytes = text.shape[1] * 4

    def occurs(seq):
        pat = np.asarray(seq, np.int32).tobytes()
        pos = raw.find(pat)
        while pos >= 0:
            if pos % 4 == 0 and pos % row_bytes + len(pat) <= row_bytes:
                return True
            pos = raw.find(pat, pos + 1)
        return False

    checked = 0
    for b in np.random.default_rng(32).integers(0, B, 300):
        ln = int(cur[b, 1])
        seq = toks[:, b].tolist()
        assert 0 <= ln <= T
        # every vocabulary id is also a one-token document (tools/gen_sam_alpaca_sam_only.py:43-44), so a match of length 1
        # always exists and needs no witness in the long documents
        assert ln >= 1
        if ln >= 2:
            assert occurs(seq[T - ln:]), (b, ln)
        if ln < T:
            assert not occurs(seq[T - ln - 1:]), (b, ln)
        checked += 1
    assert checked == 300


def test_full_size_image_round_trip(corpus, tmp_path):
    flat, off, docs, sam = corpus
    p = str(tmp_path / "full.samd")
    sam.save(p)
    again = samd_hip.StaticAutomaton.load(p).upload()
    assert again.info()["n_states"] == sam.info()["n_states"] and again.info()["n_edges"] == sam.info()["n_edges"]
    toks = _streams(docs, np.random.default_rng(41), 0.10)[:, :4096]
    a, _ = _walk(sam, np.ascontiguousarray(toks))
    b, _ = _walk(again, np.ascontiguousarray(toks))
    assert torch.equal(a, b)


def test_full_size_draft_model_matches_oracle(corpus):
    """the bs=1 request path over the full automaton: every lookup (dyn-vs-static choice, sequence / tree draft, parents,
    buffers) of two bench requests equals the oracle's, with the oracle's accepted tokens fed back to both."""
    from samd_hip import Params
    flat, off, docs, sam = corpus
    ora_static = O.StaticSAM.build([flat[off[i]:off[i + 1]] for i in range(len(off) - 1)], bench.EOS)
    p = Params(variant=0, max_predicts=60, alpha=4.0, K=8, len_bias=0)
    od = O.DraftModel(60, 4.0, 8, 0, sam_static=ora_static)
    rng = np.random.default_rng(51)
    dev = lambda a: torch.as_tensor(np.asarray(a, np.int32)).cuda()
    kinds = [0, 0]
    for _ in range(2):
        prompt, target = bench.synth_request(rng, docs)
        sess = samd_hip.Session(4096)
        od.reset(); od.update(prompt)
        d_prompt = dev(prompt)
        sess.add_tokens(d_prompt); sess.static_walk(sam, d_prompt, len(prompt), commit=True)
        pos = len(prompt)
        for step in range(150):
            start = int(target[pos])
            ty, toks, anc = od.lookup_raw(start)
            sess.draft(sam, p, dev([start]))
            d = sess.read_draft()
            assert (d.type, list(d.tokens[:d.n]), list(d.parent[:d.n])) == (ty, toks, anc), step
            if ty == 1:
                want = O.gen_buffers(anc)
                got = np.asarray(d.retrieve[:d.n_leaves * d.max_depth]).reshape(d.n_leaves, d.max_depth)
                assert got.tolist() == want["tree_retrieve_indices"].tolist()
            kinds[ty] += 1
            # scripted greedy verdict: longest root->node path that follows the continuation
            depth, ok, best = [0] * len(toks), [True] * len(toks), 0
            for i in range(1, len(toks)):
                depth[i] = depth[anc[i]] + 1
                ok[i] = ok[anc[i]] and toks[i] == target[pos + depth[i]]
                if ok[i] and depth[i] > depth[best]:
                    best = i
            a = depth[best] + 1
            acc = [int(t) for t in target[pos:pos + a]]
            pos += a
            od.update(acc)
            d_acc = dev(acc)
            sess.add_tokens(d_acc); sess.static_walk(sam, d_acc, a, commit=True)
    assert min(kinds) > 3, kinds
