"""GPU parity of the suffix-automaton kernels (through the C ABI) against the CPU oracle and the
golden fixtures produced by the reference.  Bit-exact: everything here is integer work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import samd_hip
from samd_hip import Params
from oracle import sam_oracle as O
from scripted_lm import ScriptedLM
from util import walk_visited, markov_stream, split_edges, random_parents


def dev(a, dtype=torch.int32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def so_params(max_predicts=60, alpha=4.0, K=8, len_bias=0):
    return Params(variant=0, max_predicts=max_predicts, alpha=alpha, K=K, len_bias=len_bias)


def s_params(n_predicts=40, len_threshold=5, len_bias=5):
    return Params(variant=1, max_predicts=64, alpha=4.0, K=8, len_bias=len_bias, n_predicts=n_predicts,
                  len_threshold=len_threshold)


def draft_tuple(d):
    n, nl, md = d.n, d.n_leaves, d.max_depth
    ret = np.asarray(d.retrieve[:nl * md]).reshape(nl, md).tolist()
    mask = [[(d.mask[i] >> j) & 1 for j in range(n)] for i in range(n)]
    return d.type, list(d.tokens[:n]), list(d.parent[:n]), list(d.position[:n]), mask, ret


# --------------------------------------------------------------------------------------------------
# static walk (batched, lane per stream)
# --------------------------------------------------------------------------------------------------
def check_stream_major_walk(prod, toks_tb, want_trace, want_cursors, want_visited=None, start=None):
    """samd_static_walk_streams (stream-major tokens, decoupled lanes, cursors handed out inside a wave) must give the trace, the
    final cursors and the visited-state count of the time-major kernel; lookup mode must leave the cursors alone."""
    T, B = toks_tb.shape
    toks_bt = torch.as_tensor(np.ascontiguousarray(np.asarray(toks_tb).T)).cuda()
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda") if start is None else start.clone()
    trace = torch.full((B, T, 2), -7, dtype=torch.int32, device="cuda")
    prod.walk_streams(cur, toks_bt, commit=True, trace=trace)
    assert torch.equal(trace.permute(1, 0, 2).cpu(), torch.as_tensor(np.asarray(want_trace)).to(torch.int32)), "trace"
    assert torch.equal(cur.cpu(), want_cursors.cpu()), "final cursors"
    cur2 = torch.zeros((B, 2), dtype=torch.int32, device="cuda") if start is None else start.clone()
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk_streams(cur2, toks_bt, commit=False, visited=visited)
    assert cur2.abs().sum().item() == 0 if start is None else torch.equal(cur2, start)
    if want_visited is not None:
        assert int(visited.item()) == want_visited


def test_static_walk_golden(golden):
    for case in golden("sam_traces.json.gz")["static_so"]:
        sam = samd_hip.StaticAutomaton.build(case["docs"], case["eos"], 0).upload()
        q = case["query"]
        cur = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
        trace = torch.zeros((len(q), 1, 2), dtype=torch.int32, device="cuda")
        sam.walk(cur, dev(q).reshape(-1, 1), commit=True, trace=trace)
        assert trace[:, 0].cpu().tolist() == case["walk"], case["name"]
        assert cur[0].cpu().tolist() == case["walk"][-1]
        check_stream_major_walk(sam, np.asarray(q, dtype=np.int32).reshape(-1, 1), trace.cpu().numpy(), cur)


@pytest.mark.parametrize("vocab,B,T", [(6, 300, 40), (200, 1000, 64), (5000, 4096, 32), (70000, 2048, 32), (200, 777, 13), (5000, 70000, 16)])   # >= 65535: 4-token chain words
def test_static_walk_batched_vs_oracle(vocab, B, T):
    rng = np.random.default_rng(vocab)
    docs = [markov_stream(rng, 300, vocab=vocab) for _ in range(30)] + [[i] for i in range(vocab)]
    flat = np.concatenate([np.asarray(d) for d in docs[:30]])
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    ora = O.StaticSAM.build(docs, 2)
    toks = np.empty((T, B), np.int32)
    for b in range(B):
        p = int(rng.integers(0, len(flat) - T))
        seq = flat[p:p + T].copy()
        noise = rng.random(T) < 0.15
        seq[noise] = rng.integers(0, vocab + 5, int(noise.sum()))       # incl. out-of-vocabulary ids
        toks[:, b] = seq
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    cur2 = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur2, dev(toks), commit=True, visited=visited)
    assert torch.equal(cur, cur2) and int(visited.item()) >= B * T
    if B * T <= 70000:                  # the visited-state count (the bench's algorithmic bytes) is the reference's, state for state
        assert int(visited.item()) == walk_visited(ora.export(), toks)
    got = trace.cpu().numpy()
    for b in range(0, B, max(1, B // 64)):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
            assert (int(got[t, b, 0]), int(got[t, b, 1])) == (i, l), (b, t)
    # lookup (commit=0) leaves the cursors alone
    before = cur.clone()
    prod.walk(cur, dev(toks), commit=False)
    assert torch.equal(cur, before)
    # the out-of-place lookup (static_sam.py:122-125 RETURNS the pair): same result and visited count as the committed walk, cursors untouched;
    # from the root and from the non-root cursors a first pass left
    root = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    res = torch.full((B, 2), -7, dtype=torch.int32, device="cuda")
    v2 = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.lookup_batch(root, dev(toks), res, visited=v2)
    assert torch.equal(res, cur2) and root.abs().sum().item() == 0 and int(v2.item()) == int(visited.item())
    rev = dev(toks[::-1].copy())
    want = cur.clone()
    prod.walk(want, rev, commit=True)
    prod.lookup_batch(cur, rev, res)
    assert torch.equal(res, want) and torch.equal(cur, before)
    prod.lookup_batch(cur, rev[:1], res)                                     # T = 1: the reference's lookup(token)
    for b in range(0, B, max(1, B // 32)):
        assert tuple(res[b].tolist()) == ora.transfer_state(int(before[b, 0]), int(before[b, 1]), int(toks[T - 1, b]))
    # the stream-major kernel: same trace, cursors and visited-state count; also from non-root start cursors (a second pass)
    check_stream_major_walk(prod, toks, got, cur, int(visited.item()))
    trace2 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    cur3 = cur.clone()
    prod.walk(cur3, dev(toks[::-1].copy()), commit=True, trace=trace2)
    check_stream_major_walk(prod, toks[::-1].copy(), trace2.cpu().numpy(), cur3, start=cur)


@pytest.mark.parametrize("base", [3, 66000])
def test_static_walk_long_runs_use_chain_words(base):
    """a document of all-distinct tokens is one non-branching run: a copied query is served from chain words (8 u16 / 4 u32 tokens
    per 16-byte load, csrc/samd_common.h) with reloads at word boundaries; mismatches in the middle of a word, at its first and
    last token, and ids that collide with the word's terminator must fall back to the nodes.  Every (index, length) vs the oracle."""
    doc = list(range(base, base + 400))
    docs = [doc, doc[100:180] + doc[20:60], [base + 7, base + 9, base + 8]]
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    ora = O.StaticSAM.build(docs, 2)
    rng = np.random.default_rng(base)
    streams = []
    for off in range(0, 24):
        q = doc[5 + off:5 + off + 70]
        for bad in (off % 9, 8 + off % 11, 31, 32 + off % 7):          # mismatches at assorted phases of the 8- / 4-token words
            q[bad] = [base + 399, 65535, 0xFFFF_FFFF - (1 << 32), -5, base + 1000][(off + bad) % 5]
        streams.append(q)
    toks = np.array(streams, dtype=np.int64).astype(np.int32).T.copy()      # [T, B]
    T, B = toks.shape
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    got = trace.cpu().numpy()
    for b in range(B):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
            assert (int(got[t, b, 0]), int(got[t, b, 1])) == (i, l), (b, t)
    check_stream_major_walk(prod, toks, got, cur)


@pytest.mark.parametrize("vocab", [1000, 32767, 32768, 90000])
def test_static_walk_known_climbs(vocab):
    """round 4 (csrc/sam_device.h, csrc/samd_common.h BIGRAM TABLE): a chain entry whose top bit is clear says that its state has ONE edge and
    that its suffix link is a child of the root; a cursor holding the word resolves a mismatch there with one look-up of (previous token,
    token) in the bigram table -- without loading the state or its link.  Exercised here: climbs that end at the root (miss), climbs
    whose look-up HITS (the offending token does follow the previous token elsewhere), root children of every degree (1 .. 13 edges: all
    of them live in the table), mismatches on the first / a middle / the last entry of a word and right after a half word from the table,
    token ids at the top of the vocabulary (32766 in 15-bit entries; vocab 32768 and 90000 take the 31-bit form), and the visited-state
    count.
    Every (index, length) against the oracle."""
    rng = np.random.default_rng(vocab)
    ids = rng.permutation(np.arange(3, vocab - 2))[:400].tolist()
    A = [vocab - 1, vocab - 2] + ids[:300]                     # one long run of distinct tokens, the two largest ids first
    extra = ids[300:]
    hub, hub2 = A[77], A[150]
    docs = [A, A[100:180] + A[20:60]]
    docs += [[hub, z] for z in extra[:12]]                     # state(hub) gets 13 successors: a hashed block
    docs += [[hub2, z, z + 1 if z + 1 < vocab else 5] for z in extra[12:20]]
    docs += [[t] for t in extra[20:40]]
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    ora = O.StaticSAM.build(docs, 2)
    streams = []
    for off in range(40):
        q = A[max(0, 30 + off):max(0, 30 + off) + 60]
        q[17 + off % 9] = extra[20 + off % 20]                  # a token that exists but follows nothing here: climb to the root
        q[40 + off % 5] = int(rng.integers(vocab, vocab + 3))   # out of vocabulary
        streams.append(q)
    for j, z in enumerate(extra[:12]):                          # the climb's probe hits: ... A[77] z
        streams.append((A[40 + j:78] + [z] + A[79:101 + j])[:60])
    for j, z in enumerate(extra[12:20]):                        # ... A[150] z z+1 A[153] ...
        streams.append((A[120 + j:151] + [z, z + 1 if z + 1 < vocab else 5] + A[153:190])[:60])
    for j in range(8):                                          # mismatch right after a landing through the hash's half word
        streams.append(([A[76], hub, extra[j]] + A[200 + j:257 + j])[:60])
        streams.append((A[0:3 + j] + [A[1]] + A[3 + j:70])[:60])   # the largest ids, early mismatch positions
    toks = np.array([s + [A[5]] * (60 - len(s)) for s in streams], dtype=np.int64).astype(np.int32).T.copy()
    T, B = toks.shape
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    prod.walk(torch.zeros_like(cur), dev(toks), commit=False, visited=visited)
    got = trace.cpu().numpy()
    for b in range(B):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
            assert (int(got[t, b, 0]), int(got[t, b, 1])) == (i, l), (b, t)
    assert int(visited.item()) == walk_visited(ora.export(), toks)


@pytest.mark.parametrize("base", [10, 40000])
def test_static_walk_climb_into_long_root_child(base):
    """the bigram table (csrc/samd_common.h) keeps min(length[child of a] - 1, 3) beside every pair of a: what a climbing cursor's match
    length becomes at that child.  Here token b is always preceded by the same run, so its root child holds strings of length 1 .. k + 1
    (k = 1, 2, 3, 5: the two-bit field and its escape to root16[a].w); a cursor that follows `x ... b` in registers and then meets a token
    that follows b only in another document climbs from a one-edge state straight into that child (reference: static_sam.py:98-107).
    Both entry forms (vocabulary <= 32767 and above), every (index, length) and the visited-state count against the oracle, and the
    same streams through the session's committed walk (st_transfer_tokens: resolved cursor)."""
    docs, streams = [], []
    t = base
    for k in (1, 2, 3, 5):
        run = list(range(t, t + k)); b = t + k; x, y, c, d = t + k + 1, t + k + 2, t + k + 3, t + k + 4
        tail_c = list(range(t + k + 5, t + k + 15)); tail_d = list(range(t + k + 15, t + k + 25))
        t += k + 30
        docs += [[x] + run + [b, c] + tail_c, [y] + run + [b, d] + tail_d]
        streams.append([x] + run + [b, d] + tail_d[:6])                       # climb, probe hits: length k + 1, then + 1
        streams.append([y] + run + [b, c] + tail_c[:6])
        streams.append([x] + run + [b, tail_c[3]] + tail_c[4:9])               # climb, probe misses, lands on another child
        streams.append(run + [b, d, tail_d[0], c])                             # from the root through the child of run[0]
    vocab = t + 5
    docs += [[z] for z in range(base, vocab, 7)]
    n_max = max(len(q) for q in streams)
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    ora = O.StaticSAM.build(docs, 2)
    toks = np.array([q + [docs[0][0]] * (n_max - len(q)) for q in streams], dtype=np.int32).T.copy()
    T, B = toks.shape
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    prod.walk(torch.zeros_like(cur), dev(toks), commit=False, visited=visited)
    got, fin = trace.cpu().numpy(), cur.cpu().numpy()
    lengths = ora.export()["length"]
    seen_long = 0
    for b in range(B):
        i, l = 0, 0
        for tt in range(T):
            i, l = ora.transfer_state(i, l, int(toks[tt, b]))
            assert (int(got[tt, b, 0]), int(got[tt, b, 1])) == (i, l), (b, tt)
        assert (int(fin[b, 0]), int(fin[b, 1])) == (i, l)
    for k_i, k in enumerate((1, 2, 3, 5)):                                     # the climb did land with the child's own length
        q = streams[4 * k_i]
        assert int(got[k + 2, 4 * k_i, 1]) == k + 2, (k, got[:, 4 * k_i, 1])
        seen_long += int(k + 1 > 3)                                            # child length >= 4: the field's escape value
    assert seen_long == 2
    assert int(visited.item()) == walk_visited(ora.export(), toks)
    # the single-wavefront path (st_transfer_tokens): committed cursor after each stream
    sess = samd_hip.Session(256)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for b in range(B):
        sess.reset()
        sess.static_walk(prod, dev(toks[:, b].copy()), T, commit=True, d_out=out)
        assert out.cpu().tolist() == [int(got[T - 1, b, 0]), int(got[T - 1, b, 1])], b
    check_stream_major_walk(prod, toks, got, cur, int(visited.item()))
    # the single-wavefront form (the session kernel's committed transfer: st_transfer_tokens) walks the same streams
    sess = samd_hip.Session(256)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for b in range(0, B, 3):
        sess.reset()
        sess.static_walk(prod, dev(toks[:, b].copy()), T, commit=True, d_out=out)
        assert out.cpu().tolist() == [int(got[T - 1, b, 0]), int(got[T - 1, b, 1])], b


@pytest.mark.parametrize("seed", range(10))
def test_static_walk_random_corpora(seed):
    """random small corpora of every shape the bigram table has to get right (vocabularies from 40 ids to 70000 -- both entry forms --,
    documents made of shared segments so that root children range from one edge to dozens, tokens that always follow the same run),
    random streams: corpus copies with noise, unseen in-vocabulary ids, out-of-vocabulary and negative ids, restarts from non-root
    cursors of an earlier launch.  Every (index, length), the committed cursors and the visited-state count against the oracle
    (transfer_state, static_sam.py:98-107), the stream-major kernel and the session's single-wavefront walk on the same streams."""
    rng = np.random.default_rng(1000 + seed)
    vocab = [40, 700, 5000, 32767, 32768, 70000][seed % 6]
    ids = rng.choice(np.arange(3, vocab), size=min(vocab - 3, 300), replace=False)
    segs = [rng.choice(ids, size=int(rng.integers(2, 12))).tolist() for _ in range(30)]
    docs = []
    for _ in range(int(rng.integers(8, 40))):
        d = []
        for _ in range(int(rng.integers(1, 6))):
            d += segs[int(rng.integers(0, len(segs)))]
            if rng.random() < 0.3:
                d.append(int(rng.choice(ids)))
        docs.append(d)
    docs += [[int(t)] for t in ids[::5]]
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    ora = O.StaticSAM.build(docs, 2)
    T, B = 24, 96
    toks = np.zeros((T, B), dtype=np.int64)
    for b in range(B):
        q = []
        while len(q) < T:
            d = docs[int(rng.integers(0, len(docs)))]
            a = int(rng.integers(0, len(d)))
            q += d[a:a + int(rng.integers(1, 10))]
            r = rng.random()
            if r < 0.25:
                q.append(int(rng.choice(ids)))                       # a token of the corpus, out of place
            elif r < 0.35:
                q.append(int(rng.integers(3, vocab)))                # possibly never seen
            elif r < 0.40:
                q.append(int(rng.integers(vocab, vocab + 5)))        # out of vocabulary
            elif r < 0.43:
                q.append(-1)
        toks[:, b] = q[:T]
    toks = toks.astype(np.int32)
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    prod.walk(torch.zeros_like(cur), dev(toks), commit=False, visited=visited)
    got, fin = trace.cpu().numpy(), cur.cpu().numpy()
    want_fin = np.zeros((B, 2), dtype=np.int64)
    for b in range(B):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
            assert (int(got[t, b, 0]), int(got[t, b, 1])) == (i, l), (seed, b, t)
        want_fin[b] = (i, l)
    assert (fin == want_fin).all()
    assert int(visited.item()) == walk_visited(ora.export(), toks)
    check_stream_major_walk(prod, toks, got, cur, int(visited.item()))
    # a second launch continues from the committed cursors (states of any depth, root children among them)
    toks2 = toks[::-1].copy()
    trace2 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur, dev(toks2), commit=True, trace=trace2)
    got2 = trace2.cpu().numpy()
    for b in range(0, B, 2):
        i, l = int(want_fin[b, 0]), int(want_fin[b, 1])
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks2[t, b]))
            assert (int(got2[t, b, 0]), int(got2[t, b, 1])) == (i, l), (seed, b, t)
    sess = samd_hip.Session(256)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for b in range(0, B, 8):
        sess.reset()
        sess.static_walk(prod, dev(toks[:, b].copy()), T, commit=True, d_out=out)
        assert out.cpu().tolist() == [int(want_fin[b, 0]), int(want_fin[b, 1])], (seed, b)


@pytest.mark.parametrize("vocab", [300, 70000])
def test_bigram_table_size_changes_nothing_but_bytes(vocab):
    """samd_static_set_bigram_slots: the table is an accelerator -- every size (and the default, 4 slots per root-child edge since round 5)
    gives the same cursors, traces and visited-state counts; only derived_info's bytes move."""
    rng = np.random.default_rng(5)
    docs = [markov_stream(rng, 400, vocab=vocab) for _ in range(20)] + [[i] for i in range(vocab)]
    flat = np.concatenate([np.asarray(d) for d in docs[:20]])
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    B, T = 3000, 24
    toks = np.empty((T, B), np.int32)
    for b in range(B):
        p = int(rng.integers(0, len(flat) - T))
        seq = flat[p:p + T].copy()
        noise = rng.random(T) < 0.2
        seq[noise] = rng.integers(0, vocab, int(noise.sum()))
        toks[:, b] = seq
    d_toks, root = dev(toks), torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    base = prod.derived_info()
    pairs_x4 = base["bigram_slots"]
    assert pairs_x4 >= 1024
    want, want_v = None, None
    for per_pair in (0, 2, 16, 64, 4):
        prod.set_bigram_slots(per_pair)
        info = prod.derived_info()
        if per_pair in (0, 4):
            assert info["bigram_slots"] == pairs_x4                           # 0 = the default = 4 per pair
        elif per_pair == 2:
            assert info["bigram_slots"] * 2 == pairs_x4 or info["bigram_slots"] == 1024
        else:
            assert info["bigram_slots"] == pairs_x4 * per_pair // 4
        assert info["resident_bytes"] == (prod.info()["device_bytes"] + info["chain_bytes"] + info["bigram_bytes"] + info["topk_count_bytes"] + info["edge_table_bytes"]
                                          + info["hot_word_bytes"] + info["edge_block_bytes"])
        assert (info["edge_block_slots"] > 0) != (info["edge_table_slots"] > 0) or info["edge_block_states"] == 0       # blocks REPLACE the edge table
        res = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
        v = torch.zeros(1, dtype=torch.int64, device="cuda")
        prod.lookup_batch(root, d_toks, res, visited=v)
        if want is None:
            want, want_v = res.clone(), int(v.item())
        assert torch.equal(res, want) and int(v.item()) == want_v, per_pair
    ora = O.StaticSAM.build(docs, 2)
    for b in range(0, B, 100):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
        assert tuple(want[b].tolist()) == (i, l)


@pytest.mark.parametrize("vocab,n_tok", [(300, 1 << 14), (2000, 1 << 16), (70000, 1 << 15)])
@pytest.mark.parametrize("lanes", ["lockstep", "decoupled"])
def test_static_walk_zipf_corpus_with_deep_hubs(vocab, n_tok, lanes, monkeypatch):
    """a natural-language-like corpus (bench.synth_corpus_zipf: Zipfian vocabulary, hubs of degree >> 5 at depth 1-4): the EDGE TABLE path of
    the walk (round 5: one probe per transition out of a branching state, node word 0 + probe together on every hop of a climb) against the
    oracle -- every (index, length) of every stream, the visited-state count state for state, cursors carried over a second pass -- and
    against the same automaton uploaded WITHOUT the table (SAMD_EDGE_TABLE=0).  Round 6: the default upload walks through EDGE BLOCKS and
    HOT WORDS (every slot of a branching state's block carries its fail header: one request per hop of a climb); the same automaton is
    uploaded with SAMD_EDGE_BLOCKS=0 (the round-5 edge table) and with both switched off, and all three must agree in everything."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # lanes = "decoupled": the batched launches of a handle with blocks run k_static_walk_async (every lane its own token index; an experiment,
    # off by default, SAMD_WALK_ASYNC=1) -- the same traces, cursors and visited-state counts
    monkeypatch.setenv("SAMD_WALK_ASYNC", "1" if lanes == "decoupled" else "0")
    flat, off, docs2d = bench.synth_corpus_zipf(n_tok, vocab=vocab, doc_len=128, max_succ=256)
    docs = [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    ora = O.StaticSAM.build(docs, 2)
    prod = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    info = prod.derived_info()
    assert info["edge_table_slots"] == 0 and info["edge_block_states"] > 20 and info["edge_block_bytes"] == 16 * info["edge_block_slots"]
    assert info["edge_block_slots"] >= 4 * info["edge_block_states"] and info["hot_word_bytes"] == 16 * prod.info()["n_states"]
    e = ora.export()
    assert int(((e["deg"] > 5) & (e["length"] >= 2)).sum()) > 20                    # hubs below the root children exist
    rng = np.random.default_rng(vocab)
    B, T = 3000, 32
    cdf = bench.zipf_cdf(vocab)
    d = rng.integers(0, docs2d.shape[0], B); s0 = rng.integers(0, docs2d.shape[1] - T, B)
    toks = docs2d[d[None, :], s0[None, :] + np.arange(T)[:, None]]
    noise = rng.random((T, B)) < 0.12
    ntok = (3 + np.searchsorted(cdf, rng.random((T, B)), side="right")).clip(3, vocab + 3)      # incl. out-of-vocabulary ids
    toks = np.ascontiguousarray(np.where(noise, ntok, toks).astype(np.int32))
    cur = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    trace = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    visited = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.walk(cur, dev(toks), commit=True, trace=trace)
    res = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    prod.lookup_batch(torch.zeros((B, 2), dtype=torch.int32, device="cuda"), dev(toks), res, visited=visited)
    assert torch.equal(res, cur)
    sub = toks[:, :700]
    assert int(visited.item()) >= B * T
    v2 = torch.zeros(1, dtype=torch.int64, device="cuda")
    prod.lookup_batch(torch.zeros((700, 2), dtype=torch.int32, device="cuda"), dev(np.ascontiguousarray(sub)), torch.zeros((700, 2), dtype=torch.int32, device="cuda"), visited=v2)
    assert int(v2.item()) == walk_visited(e, sub)
    got = trace.cpu().numpy()
    for b in range(0, B, 25):
        i, l = 0, 0
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[t, b]))
            assert (int(got[t, b, 0]), int(got[t, b, 1])) == (i, l), (b, t)
    # a second pass from the cursors the first one left (deep states, hubs, root children)
    rev = dev(np.ascontiguousarray(toks[::-1]))
    cur2 = cur.clone()
    trace2 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur2, rev, commit=True, trace=trace2)
    got2 = trace2.cpu().numpy()
    start = cur.cpu().numpy()
    for b in range(0, B, 40):
        i, l = int(start[b, 0]), int(start[b, 1])
        for t in range(T):
            i, l = ora.transfer_state(i, l, int(toks[T - 1 - t, b]))
            assert (int(got2[t, b, 0]), int(got2[t, b, 1])) == (i, l), (b, t)
    # the single-cursor kernels (a session's committed walk) take the same path
    sess = samd_hip.Session(64)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for b in range(0, B, 300):
        sess.reset()
        sess.static_walk(prod, dev(toks[:, b].copy()), T, commit=True, d_out=out)
        assert out.cpu().tolist() == [int(got[T - 1, b, 0]), int(got[T - 1, b, 1])]
    # the round-5 edge table instead of the blocks, then neither: identical traces, cursors and counts
    for blocks_env, table_env in (("0", "1"), ("0", "0")):
        monkeypatch.setenv("SAMD_EDGE_BLOCKS", blocks_env)
        monkeypatch.setenv("SAMD_EDGE_TABLE", table_env)
        plain = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
        pinfo = plain.derived_info()
        assert pinfo["edge_block_slots"] == 0 and pinfo["hot_word_bytes"] == 0 and (pinfo["edge_table_slots"] >= 1024) == (table_env == "1")
        cur3 = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
        trace3 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
        v3 = torch.zeros(1, dtype=torch.int64, device="cuda")
        plain.walk(cur3, dev(toks), commit=True, trace=trace3)
        plain.lookup_batch(torch.zeros((B, 2), dtype=torch.int32, device="cuda"), dev(toks), res, visited=v3)
        assert torch.equal(trace3, trace) and torch.equal(cur3, cur) and int(v3.item()) == int(visited.item()), (blocks_env, table_env)
        cur4 = cur.clone()
        trace4 = torch.zeros((T, B, 2), dtype=torch.int32, device="cuda")
        plain.walk(cur4, rev, commit=True, trace=trace4)
        assert torch.equal(trace4, trace2) and torch.equal(cur4, cur2)


def test_static_walk_empty_and_ragged():
    prod = samd_hip.StaticAutomaton.build([[3, 4, 5, 3, 4, 6]], 2, 0).upload()
    cur = torch.zeros((0, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur, torch.zeros((4, 0), dtype=torch.int32, device="cuda"))
    cur = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    prod.walk(cur, torch.zeros((0, 3), dtype=torch.int32, device="cuda"))
    assert cur.abs().sum().item() == 0
    # negative / huge token ids fall back to the root like an absent key
    prod.walk(cur, dev([[3, -1, 10 ** 9], [4, 3, 3]]))
    assert cur.cpu().tolist() == [[2, 2], [1, 1], [1, 1]]
    # the out-of-place lookup: empty batch; no tokens = the cursors themselves
    prod.lookup_batch(torch.zeros((0, 2), dtype=torch.int32, device="cuda"), torch.zeros((4, 0), dtype=torch.int32, device="cuda"),
                      torch.zeros((0, 2), dtype=torch.int32, device="cuda"))
    res = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    prod.lookup_batch(cur, torch.zeros((0, 3), dtype=torch.int32, device="cuda"), res)
    assert res.cpu().tolist() == [[2, 2], [1, 1], [1, 1]]
    # the stream-major entry point: empty batch, no tokens, the same three streams
    prod.walk_streams(torch.zeros((0, 2), dtype=torch.int32, device="cuda"), torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
    cur = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    prod.walk_streams(cur, torch.zeros((3, 0), dtype=torch.int32, device="cuda"))
    assert cur.abs().sum().item() == 0
    prod.walk_streams(cur, dev([[3, 4], [-1, 3], [10 ** 9, 3]]))
    assert cur.cpu().tolist() == [[2, 2], [1, 1], [1, 1]]


# --------------------------------------------------------------------------------------------------
# dynamic automaton
# --------------------------------------------------------------------------------------------------
def check_dyn(sess, ora):
    got, want = sess.export(), ora.export()
    assert got["error"] == 0
    for k in ("link", "length", "aux", "deg", "edge_tok", "edge_dst", "text"):       # edges in dict order
        assert np.array_equal(got[k], want[k]), k
    assert (got["cur_index"], got["cur_length"]) == ora.cursor()
    assert (got["last"], got["max_length"]) == (ora.last, ora.max_length)


def test_dyn_golden(golden):
    for case in golden("sam_traces.json.gz")["dyn"]:
        sess = samd_hip.Session(len(case["tokens"]) + 8)
        ora = O.DynSAM()
        toks, cuts = case["tokens"], case["cuts"]
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        for (a, b), cur, probes in zip(zip(cuts[:-1], cuts[1:]), case["cursors"], case["probes"]):
            sess.add_tokens(dev(toks[a:b]))
            ora.add_tokens(toks[a:b])
            for t, pi, pl in probes:
                sess.dyn_walk(dev([t]), 1, commit=False, d_out=out)
                assert out.cpu().tolist() == [pi, pl]
        check_dyn(sess, ora)
        g = sess.export()
        assert g["text"].tolist() == case["input_ids"]
        assert g["link"].tolist() == case["table"]["link"] and g["aux"].tolist() == case["table"]["aux"]
        assert [[list(e) for e in st] for st in split_edges(g)] == case["table"]["edges"]


@pytest.mark.parametrize("vocab,n", [(3, 600), (5, 2048), (300, 2048), (32000, 1500)])
def test_dyn_random_streams(vocab, n):
    rng = np.random.default_rng(vocab + n)
    toks = markov_stream(rng, n, vocab=max(vocab, 6)) if vocab > 5 else rng.integers(0, vocab, n).tolist()
    sess = samd_hip.Session(n + 4)
    ora = O.DynSAM()
    pos = 0
    while pos < n:
        k = int(rng.integers(1, 64)) if pos else n // 2
        sess.add_tokens(dev(toks[pos:pos + k]))
        ora.add_tokens(toks[pos:pos + k])
        pos += k
    check_dyn(sess, ora)
    # device-side count + reset
    sess.reset()
    ora.reset()
    cnt = dev([100])
    sess.add_tokens(dev(toks[:200]), n=200, d_n=cnt)
    ora.add_tokens(toks[:100])
    check_dyn(sess, ora)


def test_dyn_capacity_is_reported_not_corrupted():
    sess = samd_hip.Session(16)
    sess.add_tokens(dev(list(range(3, 40))))
    assert sess.export(with_edges=False)["error"] == -2


# --------------------------------------------------------------------------------------------------
# drafts + buffers
# --------------------------------------------------------------------------------------------------
def test_tree_buffers(golden):
    g = golden("buffers.json.gz")
    rng = np.random.default_rng(1)
    cases = [(c["anc"], 0) for c in g["so"]]
    cases += [(random_parents(rng, n, s), r) for n in (1, 5, 40, 64, 65, 111, 128) for s in ("chain", "star", "bushy", "random") for r in (0, 1)]
    for anc, rev in cases:
        n = len(anc)
        pos = torch.zeros(n, dtype=torch.int32, device="cuda")
        mask = torch.zeros(n if n <= 64 else 2 * n, dtype=torch.int64, device="cuda")      # n > 64: the n high words follow the n low words
        mb = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
        ret = torch.full((n * n,), -7, dtype=torch.int32, device="cuda")
        shape = torch.zeros(2, dtype=torch.int32, device="cuda")
        samd_hip.check(samd_hip.lib().samd_tree_buffers(samd_hip._ptr(dev(anc)), n, rev, samd_hip._ptr(pos), samd_hip._ptr(mask),
                                                        samd_hip._ptr(mb), samd_hip._ptr(ret), samd_hip._ptr(shape),
                                                        samd_hip.current_stream()))
        want = O.gen_buffers(anc)
        nl, md = shape.cpu().tolist()
        assert pos.cpu().tolist() == want["tree_position_ids"][0].tolist()
        assert mb.cpu().numpy().astype(bool).tolist() == want["tree_attn_mask"][0, 0].tolist()
        m64 = mask.cpu().numpy().astype(np.uint64)
        bit = lambda i, j: int(m64[i] >> np.uint64(j)) & 1 if j < 64 else int(m64[n + i] >> np.uint64(j - 64)) & 1
        assert [[bit(i, j) for j in range(n)] for i in range(n)] == want["tree_attn_mask"][0, 0].astype(int).tolist()
        wr = want["tree_retrieve_indices"]
        got = ret[:nl * md].reshape(nl, md).cpu().numpy()
        assert got.tolist() == (wr[::-1] if rev else wr).tolist()
    tr = g["token_recycle"][0]
    want = O.tr_gen_buffers(tr["tree"])
    assert want["tree_retrieve_indices"].tolist() == tr["retrieve"]


def test_drafts_golden(golden):
    g = golden("drafts.json.gz")
    for case in g["dyn_so"]:
        sess = samd_hip.Session(len(case["tokens"]) + 8)
        sess.add_tokens(dev(case["tokens"]))
        p = so_params(case["max_predicts"], case["alpha"])
        for c in case["cases"]:
            sess.draft_seq(p, c["index"], c["match"], c["start"])
            ty, toks, par, pos, mask, ret = draft_tuple(sess.read_draft())
            assert ty == 0 and toks == c["seq"] and pos == c["pos"][0]
            assert par == [i - 1 for i in range(len(toks))] and ret == [list(range(len(toks)))]
    for case in g["dyn_s"]:
        sess = samd_hip.Session(len(case["tokens"]) + 8)
        sess.add_tokens(dev(case["tokens"]))
        p = s_params(case["n_predicts"])
        for c in case["cases"]:
            sess.draft_fixed(None, p, 0, c["index"], c["start"])
            assert draft_tuple(sess.read_draft())[1] == c["seq"]
    corp_s = {c["name"]: c for c in golden("sam_traces.json.gz")["static_s"]}
    corp_s.update(g.get("corpora", {}))
    for case in g["static_s"]:
        sam = samd_hip.StaticAutomaton.build(corp_s[case["name"]]["docs"], corp_s[case["name"]]["eos"], 1).upload()
        sess = samd_hip.Session(64)
        p = s_params(case["n_predicts"])
        for c in case["cases"]:
            sess.draft_fixed(sam, p, 1, c["index"], c["start"])
            assert draft_tuple(sess.read_draft())[1] == c["seq"]
    corp = {c["name"]: c for c in golden("sam_traces.json.gz")["static_so"]}
    corp.update(g.get("corpora", {}))
    n_tree = 0
    for case in g["tree"]:
        sam = samd_hip.StaticAutomaton.build(corp[case["name"]]["docs"], corp[case["name"]]["eos"], 0).upload()
        sess = samd_hip.Session(64)
        p = so_params(case["max_predicts"], case["alpha"], case["K"])
        for c in case["cases"]:
            sess.draft_tree(sam, p, c["index"], c["match"], c["start"])
            ty, toks, par, pos, mask, ret = draft_tuple(sess.read_draft())
            assert ty == 1 and toks == c["tree"] and par == c["anc"], (case["name"], c)
            want = O.gen_buffers(c["anc"])
            assert ret == want["tree_retrieve_indices"].tolist() and pos == want["tree_position_ids"][0].tolist()
            n_tree += 1
    assert n_tree > 50


def test_draft_model_golden_so(golden):
    for case in golden("draft_model.json.gz")["so"]:
        sam = samd_hip.StaticAutomaton.build(case["docs"], case["eos"], 0).upload()
        sess = samd_hip.Session(512)
        p = so_params(case["max_predicts"], case["alpha"], case["K"], case["len_bias"])
        sess.add_tokens(dev(case["prompt"]))
        sess.static_walk(sam, dev(case["prompt"]), len(case["prompt"]), commit=True)
        for s in case["steps"]:
            d_start = dev([s["start"]])
            sess.draft(sam, p, d_start)
            ty, toks, par, pos, mask, ret = draft_tuple(sess.read_draft())
            assert ("sequence", "tree")[ty] == s["type"] and toks == s["tokens"]
            if ty == 1:
                assert [pos] == s["pos"] and ret == s["retrieve"] and mask == s["mask"]
            acc = dev(s["accepted"])
            sess.add_tokens(acc)
            sess.static_walk(sam, acc, len(s["accepted"]), commit=True)
            e = sess.export(with_edges=False)
            assert [e["cur_index"], e["cur_length"], e["st_index"], e["st_length"]] == s["cursors"]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_draft_model_random_vs_oracle(seed):
    """bigger automata, long drafts (max_predicts 60), ties in counts: product DraftModel.lookup == oracle."""
    rng = np.random.default_rng(seed)
    V = 64
    docs = [markov_stream(rng, 400, vocab=V, succ=3, noise=0.03) for _ in range(40)] + [[i] for i in range(V)]
    flat = [t for d in docs[:40] for t in d]
    sam = samd_hip.StaticAutomaton.build(docs, 2, 0).upload()
    st = O.StaticSAM.build(docs, 2)
    od = O.DraftModel(60, 4.0, 8, int(seed), sam_static=st)
    p = so_params(60, 4.0, 8, int(seed))
    sess = samd_hip.Session(2048)
    q = int(rng.integers(0, len(flat) - 1200))
    prompt = flat[q:q + 120] + flat[q + 30:q + 90]
    od.reset(); od.update(prompt)
    d_prompt = dev(prompt)
    sess.add_tokens(d_prompt); sess.static_walk(sam, d_prompt, len(prompt), commit=True)
    # the "true" text that gets committed: fresh corpus spans (static matches) and prompt repeats (dyn matches)
    text = []
    while len(text) < 700:
        if rng.random() < 0.6:
            r = int(rng.integers(0, len(flat) - 80)); text += flat[r:r + int(rng.integers(8, 60))]
        else:
            r = int(rng.integers(0, len(prompt) - 20)); text += prompt[r:r + int(rng.integers(4, 20))]
        if rng.random() < 0.3:
            text += rng.integers(3, V, 2).tolist()
    kinds, pos = [0, 0], 0
    for step in range(120):
        start = text[pos]
        ty, toks, anc = od.lookup_raw(start)
        d_start = dev([start])
        sess.draft(sam, p, d_start)
        d = sess.read_draft()
        assert (d.type, list(d.tokens[:d.n]), list(d.parent[:d.n])) == (ty, toks, anc), step
        want = O.gen_buffers(anc)
        assert np.asarray(d.retrieve[:d.n_leaves * d.max_depth]).reshape(d.n_leaves, d.max_depth).tolist() == want["tree_retrieve_indices"].tolist()
        kinds[ty] += 1
        a = int(rng.integers(1, 6))
        acc = text[pos:pos + a]
        pos += a
        od.update(acc)
        d_acc = dev(acc)
        sess.add_tokens(d_acc); sess.static_walk(sam, d_acc, a, commit=True)
    assert min(kinds) > 5, kinds


def test_draft_model_golden_s(golden):
    tr_tree = golden("buffers.json.gz")["token_recycle"][0]["tree"]
    off = np.zeros(len(tr_tree) + 1, np.int32); off[1:] = np.cumsum([len(c) for c in tr_tree])
    ch = np.asarray([c for cs in tr_tree for c in cs], np.int32)
    for case in golden("draft_model.json.gz")["s"]:
        V = case["vocab"]
        sam = samd_hip.StaticAutomaton.build(case["docs"], case["eos"], 1).upload() if case["use_static"] else None
        sess = samd_hip.Session(512)
        p = s_params(case["n_predicts"], case["len_threshold"], case["len_bias"])
        h = samd_hip.C.c_void_p()
        samd_hip.check(samd_hip.lib().samd_recycle_create(V, samd_hip._ptr(off), samd_hip._ptr(ch), len(tr_tree), samd_hip.C.byref(h)))
        def tr_update(tokens, logits):
            lg = torch.tensor(logits, dtype=torch.float32, device="cuda")
            samd_hip.check(samd_hip.lib().samd_recycle_update(h, samd_hip._ptr(dev(tokens)), samd_hip._ptr(lg), samd_hip.F32, len(tokens),
                                                              None, V, V, samd_hip.current_stream()))
        pr = dev(case["prompt"])
        sess.add_tokens(pr)
        sess.static_walk(sam, pr, len(case["prompt"]), commit=True)
        tr_update(case["prompt"], case["prompt_logits"])
        for s in case["steps"]:
            st = dev([s["start"]])
            sess.draft(sam, p, st)
            d = sess.read_draft()
            if d.type == 2:
                out = torch.zeros(len(tr_tree), dtype=torch.int32, device="cuda")
                samd_hip.check(samd_hip.lib().samd_recycle_draft(h, samd_hip._ptr(st), samd_hip._ptr(out), samd_hip.current_stream()))
                ty, toks = "tree", out.cpu().tolist()
            else:
                ty, toks = "sequence", list(d.tokens[:d.n])
            assert ty == s["type"] and toks == s["tokens"]
            acc = dev(s["accepted"])
            sess.add_tokens(acc)
            sess.static_walk(sam, acc, len(s["accepted"]), commit=True)
            tr_update(toks, s["logits"])
        samd_hip.lib().samd_recycle_free(h)


def test_token_recycle_golden(golden):
    for case in golden("token_recycle.json.gz"):
        tree, V = case["tree"], case["vocab"]
        off = np.zeros(len(tree) + 1, np.int32); off[1:] = np.cumsum([len(c) for c in tree])
        ch = np.asarray([c for cs in tree for c in cs] or [0], np.int32)
        h = samd_hip.C.c_void_p()
        samd_hip.check(samd_hip.lib().samd_recycle_create(V, samd_hip._ptr(off), samd_hip._ptr(ch), len(tree), samd_hip.C.byref(h)))
        ora = O.TokenRecycle(tree, V)
        for r in case["rounds"]:
            for dt in (torch.float32, torch.float16):
                lg = torch.tensor(r["logits"], dtype=torch.float32, device="cuda").to(dt)
                samd_hip.check(samd_hip.lib().samd_recycle_update(h, samd_hip._ptr(dev(r["tree_tokens"])), samd_hip._ptr(lg),
                                                                  samd_hip.torch_dtype_code(dt), len(r["tree_tokens"]), None, V, V,
                                                                  samd_hip.current_stream()))
            ora.update(r["tree_tokens"], O.topk8_rows(np.asarray(r["logits"], np.float32)))
            for start, draft in r["drafts"]:
                out = torch.zeros(len(tree), dtype=torch.int32, device="cuda")
                samd_hip.check(samd_hip.lib().samd_recycle_draft(h, samd_hip._ptr(dev([start])), samd_hip._ptr(out), samd_hip.current_stream()))
                assert out.cpu().tolist() == draft
            tab = np.zeros((V, 8), np.int32); pres = np.zeros(V, np.uint8)
            samd_hip.check(samd_hip.lib().samd_recycle_export(h, samd_hip._ptr(tab), samd_hip._ptr(pres), samd_hip.current_stream()))
            assert np.array_equal(pres, ora.present) and np.array_equal(tab[pres > 0], ora.table[ora.present > 0])
        samd_hip.lib().samd_recycle_free(h)


# --------------------------------------------------------------------------------------------------
# posterior / accept
# --------------------------------------------------------------------------------------------------
def test_accept_golden(golden):
    sess = samd_hip.Session(64)
    for c in golden("posterior.json.gz"):
        logits = torch.tensor(c["logits"], dtype=torch.float32, device="cuda")
        n = len(c["tokens"])
        am = torch.zeros(n, dtype=torch.int32, device="cuda")
        samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(logits), samd_hip.F32, n, logits.shape[1], logits.shape[1], None,
                                                       samd_hip._ptr(am), samd_hip.current_stream()))
        assert am.cpu().tolist() == O.argmax_rows(np.asarray(c["logits"], np.float32)).tolist()
        sess.set_draft(dev(c["tokens"]), dev(c["anc"]), n, type_=0 if c["type"] == "sequence" else 1)
        sess.accept(am)
        v = sess.read_verdict()
        assert (v.best, v.accept) == (c["best"], c["accept"])
        assert list(v.tokens[:v.accept]) == c["accepted_tokens"]
        assert v.next_token == c["next_argmax"]
        if c["type"] == "tree":
            assert list(v.kv_index[:v.accept]) == c["accepted_indices"]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,vocab", [(1, 32000), (60, 32000), (64, 128256), (7, 1003)])
def test_argmax_rows(dtype, rows, vocab):
    g = torch.Generator(device="cuda").manual_seed(rows * vocab)
    x = torch.randn((rows, vocab), generator=g, device="cuda", dtype=torch.float32).to(dtype)
    x[0, 5] = x[0, vocab - 1] = 100.0          # exact tie: first index wins
    if rows > 1:
        x[1, :] = 0.0                           # all equal -> 0
    out = torch.full((rows,), -1, dtype=torch.int32, device="cuda")
    samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(x), samd_hip.torch_dtype_code(dtype), rows, vocab, vocab, None,
                                                   samd_hip._ptr(out), samd_hip.current_stream()))
    want = O.argmax_rows(x.float().cpu().numpy())
    assert out.cpu().tolist() == want.tolist()
    assert out[0].item() == 5
    # device-side row count
    cnt = dev([max(1, rows // 2)])
    out2 = torch.full((rows,), -1, dtype=torch.int32, device="cuda")
    samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(x), samd_hip.torch_dtype_code(dtype), rows, vocab, vocab, samd_hip._ptr(cnt),
                                                   samd_hip._ptr(out2), samd_hip.current_stream()))
    k = max(1, rows // 2)
    assert out2[:k].cpu().tolist() == want[:k].tolist() and (out2[k:] == -1).all()


# --------------------------------------------------------------------------------------------------
# whole loop: prefill + fused step kernel, scripted LM, vs the reference's recorded traces
# --------------------------------------------------------------------------------------------------
def run_device_loop(case, fused):
    lm = ScriptedLM(case["target"], case["vocab"])
    sam = samd_hip.StaticAutomaton.build(case["docs"], case["eos"], 0).upload()
    sess = samd_hip.Session(case["max_cache_len"] + 64)
    p = so_params(case["max_predicts"], case["alpha"], case["K"], case["len_bias"])
    prompt = case["prompt"]
    ids = list(prompt)
    sess.reset()
    pr = dev(prompt)
    sess.add_tokens(pr)
    sess.static_walk(sam, pr, len(prompt), commit=True)
    start = int(np.argmax(lm.logits([], prompt, [i - 1 for i in range(len(prompt))])[-1]))
    d_start = dev([start])
    sess.draft(sam, p, d_start)
    dt, ds, acc_list, trace = 0, 0, [], []
    for _ in range(case["max_new_tokens"]):
        if len(prompt) + dt + case["max_predicts"] >= case["max_cache_len"]:
            break
        d = sess.read_draft()
        toks, anc = list(d.tokens[:d.n]), list(d.parent[:d.n])
        logits = torch.from_numpy(lm.logits(ids, toks, anc)).cuda().half()
        am = torch.zeros(d.n, dtype=torch.int32, device="cuda")
        samd_hip.check(samd_hip.lib().samd_argmax_rows(samd_hip._ptr(logits), samd_hip.F16, d.n, logits.shape[1], logits.shape[1], None,
                                                       samd_hip._ptr(am), samd_hip.current_stream()))
        if fused:
            sess.step(sam, p, am)
        else:
            sess.accept(am)
            sess.commit(sam)
            sess.draft(sam, p, sess.device_views()["start_token"])
        v = sess.read_verdict()
        new = list(v.tokens[:v.accept])
        full = list(new)
        stop = False
        if case["eos"] in new:
            new = new[:new.index(case["eos"]) + 1]
            stop = True
        ids.extend(new)
        ds += 1; dt += len(new); acc_list.append(len(new))
        trace.append({"type": "sequence" if d.type == 0 else "tree", "tokens": toks, "anc": anc, "best": v.best, "accept": v.accept,
                      "accepted": full, "kv_indices": None if d.type == 0 else list(v.kv_index[:v.accept]),
                      "node_argmax": am.cpu().tolist()})
        if stop or dt >= case["max_new_tokens"]:
            break
    e = sess.export(with_edges=False)
    assert e["error"] == 0
    return {"output_ids": ids[:len(prompt) + case["max_new_tokens"]], "decode_tokens": dt, "decode_steps": ds,
            "accept_lengths": acc_list, "trace": trace}, e


@pytest.mark.parametrize("fused", [True, False])
def test_loop_so_golden(golden, fused):
    for case in golden("loop_so.json.gz"):
        got, e = run_device_loop(case, fused)
        for k in ("output_ids", "decode_tokens", "decode_steps", "accept_lengths"):
            assert got[k] == case[k], k
        for gt, wt in zip(got["trace"], case["trace"]):
            assert gt == wt
        assert got["output_ids"] == case["target"][:len(got["output_ids"])]      # lossless
        # the dynamic automaton's text is exactly prompt + every committed (untruncated) token
        committed = case["prompt"] + [t for st in case["trace"] for t in st["accepted"]]
        assert e["n_text"] - 1 == len(committed)


def test_tree_buffers_malformed_parents_terminate():
    """an externally supplied draft whose parent array is not a tree (forward reference, self loop, out of range) must not
    hang the wavefront: such nodes are re-attached to the root."""
    bad = [-1, 0, 5, 3, 99, 1, -7, 2]
    fixed = [-1, 0, 0, 0, 0, 1, 0, 2]
    n = len(bad)
    sess = samd_hip.Session(64)
    sess.set_draft(dev(list(range(10, 10 + n))), dev(bad), n, type_=1)
    d = sess.read_draft()
    want = O.gen_buffers(fixed)
    assert list(d.parent[:n]) == fixed
    assert list(d.position[:n]) == want["tree_position_ids"][0].tolist()
    ret = np.asarray(d.retrieve[:d.n_leaves * d.max_depth]).reshape(d.n_leaves, d.max_depth)
    assert ret.tolist() == want["tree_retrieve_indices"].tolist()


def test_scripted_logits_stay_inside_the_buffer():
    """samd_scripted_logits (bench / tests only) writes one ranked row per draft node; a draft with more nodes than the logits buffer
    has rows (a 61-node tree replayed through the 48-row bucket's graph, as bench.py's per-bucket breakdown does) must leave the
    memory behind the buffer alone."""
    n, rows, vocab = 61, 48, 512
    sess = samd_hip.Session(256)
    sess.set_draft(dev(list(range(10, 10 + n))), dev(random_parents(np.random.default_rng(5), n, "bushy")), n, type_=1)
    arg = dev([(7 * i + 3) % vocab for i in range(64)])
    guard = torch.zeros((64, vocab), dtype=torch.float16, device="cuda")
    sess.scripted_logits(arg, guard[:rows], vocab)
    torch.cuda.synchronize()
    assert guard[:rows].argmax(dim=1).tolist() == arg[:rows].tolist()
    assert float(guard[rows:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("vocab,rows", [(512, 7), (32000, 61), (4097, 3), (128256, 9)])
def test_recycle_update_split_topk_matches_torch(dtype, vocab, rows):
    """TokenRecycle.update (token_recycle.py:40-48) over half-precision logits takes the split top-8 (k_topk8_part / _merge): the
    table rows equal logits.topk(8).indices in (value desc, index asc) order -- half precision produces exact ties --, a token that
    occurs twice keeps its LAST row, and rows past the device-side count are ignored."""
    g = torch.Generator(device="cuda").manual_seed(vocab + rows)
    logits = (torch.randn((rows, vocab), generator=g, device="cuda") * 3).to(dtype)
    if vocab == 4097:
        logits[:, -1] = 20.0                                  # the best element sits alone in the last segment
    tokens = torch.randperm(vocab, generator=g, device="cuda")[:rows].to(torch.int32)
    if rows > 2:
        tokens[rows - 1] = tokens[0]                          # later row wins
    off = np.zeros(2, np.int32); ch = np.zeros(1, np.int32)
    h = samd_hip.C.c_void_p()
    samd_hip.check(samd_hip.lib().samd_recycle_create(vocab, samd_hip._ptr(off), samd_hip._ptr(ch), 1, samd_hip.C.byref(h)))
    n_live = torch.tensor([rows - 1 if vocab == 512 else rows], dtype=torch.int32, device="cuda")      # one case with a device-side row count
    samd_hip.check(samd_hip.lib().samd_recycle_update(h, samd_hip._ptr(tokens), samd_hip._ptr(logits), samd_hip.torch_dtype_code(dtype), rows,
                                                      samd_hip._ptr(n_live), vocab, vocab, samd_hip.current_stream()))
    table = np.zeros((vocab, 8), np.int32); present = np.zeros(vocab, np.uint8)
    samd_hip.check(samd_hip.lib().samd_recycle_export(h, samd_hip._ptr(table), samd_hip._ptr(present), samd_hip.current_stream()))
    samd_hip.lib().samd_recycle_free(h)
    want = torch.argsort(-logits.float(), dim=-1, stable=True)[:, :8].cpu().numpy()
    live = int(n_live.item())
    toks = tokens.cpu().numpy()
    expect = {}
    for r in range(live):
        expect[int(toks[r])] = want[r]
    assert int(present.sum()) == len(expect)
    for t, row in expect.items():
        assert present[t] == 1 and table[t].tolist() == row.tolist(), (t, table[t], row)
