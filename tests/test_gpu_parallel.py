"""GPU path of the request-parallel helpers with a one-rank RCCL group (the box has a single GPU): the static
automaton's image travels through device tensors and is adopted by the library (samd_static_adopt_device), results are
all-gathered on the device.  The multi-rank logic itself is covered on CPU with gloo (tests/test_parallel_cpu.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist

import samd_hip
from util import markov_stream


def test_broadcast_adopt_and_gather_on_device():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from samd_hip import parallel
        from oracle import sam_oracle as O
        rng = np.random.default_rng(2)
        docs = [markov_stream(rng, 200, vocab=80) for _ in range(10)] + [[i] for i in range(80)]
        built = samd_hip.StaticAutomaton.build(docs, 2, samd_hip.KIND_COUNT)
        auto = parallel.broadcast_static(built, src=0)               # device tensors + adopt
        assert auto.info()["uploaded"] == 1
        ora = O.StaticSAM.build(docs, 2)
        ora.reset()                                                   # build leaves the cursor where the last document ended
        toks = np.asarray(markov_stream(rng, 64, vocab=80), dtype=np.int32)
        cur = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
        trace = torch.zeros((len(toks), 1, 2), dtype=torch.int32, device="cuda")
        auto.walk(cur, torch.from_numpy(toks.reshape(-1, 1)).cuda(), commit=True, trace=trace)
        want = []
        for t in toks.tolist():
            ora.transfer_tokens([t])
            want.append(list(ora.cursor()))
        assert trace[:, 0].cpu().tolist() == want
        rows = [[1, 2, 3], [], [7]]
        assert parallel.gather_results(rows) == [rows]
        assert parallel.shard_bounds(10, 1, 0) == (0, 10)
    finally:
        dist.destroy_process_group()
