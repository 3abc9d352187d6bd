"""CPU (gloo, world size 2) coverage of the request-parallel helpers: contiguous sharding as
evaluation/eval_vicuna.py:50-65, broadcast of the static automaton's flat image from rank 0, all-gather of ragged
results.  The GPU path differs only in where the four image regions live (RCCL into device buffers + adopt)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import samd_hip
    from samd_hip import parallel
    from util import markov_stream
    try:
        rng = np.random.default_rng(0)
        docs = [markov_stream(rng, 120, vocab=60) for _ in range(8)] + [[i] for i in range(60)]
        ref = samd_hip.StaticAutomaton.build(docs, 2, samd_hip.KIND_COUNT)
        mine = parallel.broadcast_static(ref if rank == 0 else None, src=0)
        a, b = ref.export(), mine.export()
        same = all(np.array_equal(a[k], b[k]) for k in a) and ref.info()["n_spill"] == mine.info()["n_spill"]
        # shard 11 requests, every rank produces ragged rows for its chunk
        lo, hi = parallel.shard_bounds(11, world, rank)
        rows = [list(range(100 * i, 100 * i + (i % 4))) for i in range(lo, hi)]
        got = parallel.gather_results(rows)
        flat = [r for per_rank in got for r in per_rank]
        q.put((rank, same, (lo, hi), flat))
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [list(range(100 * i, 100 * i + (i % 4))) for i in range(11)]
    assert [r[2] for r in res] == [(0, 6), (6, 11)]          # chunk = 11 // 2, the leftover question goes to the first rank
    for rank, same, _, flat in res:
        assert same, "broadcast automaton differs from the source"
        assert flat == want                                   # every rank sees all results, in request order


def _eval_worker(rank, world, port, qfile, ans, q):
    for p in (ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from evaluation import run_eval
        from test_harness_cpu import ToyTokenizer
        tok = ToyTokenizer()
        seen = []

        def forward(inputs, model, tokenizer, max_new_tokens):
            n = len(inputs.input_ids[0])
            seen.append(n)
            new = [tok._id(w) for w in f"answer of rank {rank} </s>".split()]
            return [inputs.input_ids[0].tolist() + new], len(new), 2, [n % 5 + 1, 1]

        acc = run_eval(object(), tok, forward, "toy", qfile, None, None, ans, max_new_tokens=8)
        q.put((rank, len(seen), acc))
    finally:
        dist.destroy_process_group()


def test_run_eval_shards_and_gathers_world2(tmp_path):
    """run_eval under a 2-rank gloo group: contiguous question chunks (eval_vicuna.py:50-65), records and accept lengths gathered
    over the process group, rank 0 writes ONE answer file ordered by question id, no per-rank files are left behind."""
    import json
    qs = [{"question_id": 10 + i, "category": "qa", "turns": [f"question number {i}"]} for i in range(5)]
    qfile = tmp_path / "q.jsonl"
    qfile.write_text("".join(json.dumps(x) + "\n" for x in qs))
    ans = tmp_path / "out" / "a.jsonl"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, str(qfile), str(ans), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = [json.loads(l) for l in ans.read_text().splitlines()]
    assert [r["question_id"] for r in rows] == [10, 11, 12, 13, 14]
    who = [r["choices"][0]["turns"][0] for r in rows]
    assert who == ["answer of rank 0"] * 3 + ["answer of rank 1"] * 2          # chunk = 5 // 2, the leftover question to the first rank
    assert res[0][1] == 3 * 1 + 3 and res[1][1] == 3 * 1 + 2                  # 3 warm-up passes + own questions
    assert res[0][2] == res[1][2] and len(res[0][2]) == 2 * 5                 # every rank holds all accept lengths, in question order
    assert sorted(os.listdir(tmp_path / "out")) == ["a.jsonl"]


def test_run_eval_world8_with_a_question_count_that_does_not_divide(tmp_path):
    """configs[4]'s shape on CPU: 8 ranks (gloo), 21 questions = 8 x 2 + 5: the first five ranks take 3 questions, the others 2;
    every question answered exactly once, one answer file in question order, every rank holds all 42 accept lengths."""
    import json
    n_q, world = 21, 8
    qs = [{"question_id": 100 + i, "category": "qa", "turns": [f"question number {i}"]} for i in range(n_q)]
    qfile = tmp_path / "q.jsonl"
    qfile.write_text("".join(json.dumps(x) + "\n" for x in qs))
    ans = tmp_path / "out" / "a.jsonl"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, str(qfile), str(ans), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = [json.loads(l) for l in ans.read_text().splitlines()]
    assert [r["question_id"] for r in rows] == list(range(100, 100 + n_q))
    who = [r["choices"][0]["turns"][0] for r in rows]
    want = sum((["answer of rank %d" % r] * (3 if r < 5 else 2) for r in range(world)), [])
    assert who == want
    assert [r[1] for r in res] == [3 + (3 if r < 5 else 2) for r in range(world)]      # 3 warm-up passes + own questions
    assert all(r[2] == res[0][2] for r in res) and len(res[0][2]) == 2 * n_q
    assert sorted(os.listdir(tmp_path / "out")) == ["a.jsonl"]


def test_shard_bounds_cover_everything():
    from samd_hip import parallel
    for n in (0, 1, 7, 80, 481, 487):
        for world in (1, 2, 4, 8):
            spans = [parallel.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)     # balanced, the extra ones first


def test_shard_bounds_reference_tail_keeps_the_reference_cut_points(monkeypatch):
    """eval_vicuna.py:50-65: range(0, n, n // world) -- `world` chunks of n // world and one extra chunk of the remainder"""
    from samd_hip import parallel
    for n in (7, 80, 481, 487):
        for world in (1, 2, 4, 8):
            if n // world == 0:
                continue
            chunk = n // world
            ref = [(i, min(i + chunk, n)) for i in range(0, n, chunk)]                      # the reference's slices questions[i: i + chunk_size]
            spans = [parallel.shard_bounds(n, world, r, tail="reference") for r in range(world)]
            assert spans[:-1] == ref[:world - 1]
            assert spans[-1] == (ref[world - 1][0], n)                                      # the last rank: its own chunk, then the extra one
            assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    monkeypatch.setenv("SAMD_SHARD_TAIL", "reference")
    assert parallel.shard_bounds(487, 8, 7) == (420, 487) and parallel.shard_bounds(487, 8, 0) == (0, 60)
    monkeypatch.setenv("SAMD_SHARD_TAIL", "nonsense")
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 2, 0)
