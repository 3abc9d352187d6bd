"""Request-level data parallelism over the GPUs of one node.

The reference's only parallelism is Ray actors that each run the whole model on a contiguous chunk of the question
list and append to a shared jsonl (evaluation/eval_vicuna.py:39-68, :233-258).  Here: one process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI), the static automaton's flat image is broadcast once from rank 0,
every rank decodes its own request shard with its own dynamic automaton and KV cache, and the padded results are
all-gathered at the end.  Nothing on the per-step path communicates.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import StaticAutomaton


def shard_bounds(n_items, world, rank, tail=None):
    """contiguous chunks as eval_vicuna.py:50-65 (chunk = n // world questions per worker, in question order).  The reference
    hands the remainder to one extra Ray task that runs after the others; with a fixed world that tail would double the last
    rank's wall time (480 questions + 7 on 8 GPUs), so by default (tail "spread") the n % world leftover questions are spread one each
    over the first ranks: rank r gets chunk + 1 when r < n % world.  Still contiguous, still in order, sizes differ by at most one.
    tail "reference" (or SAMD_SHARD_TAIL=reference) keeps the reference's cut points instead -- `world` chunks of n // world questions
    and the extra chunk of n % world, which the LAST rank takes after its own (Ray gives it to whichever actor frees up first; the
    answers and their order in the gathered file are the same either way)."""
    import os
    tail = tail or os.environ.get("SAMD_SHARD_TAIL", "spread")
    if tail not in ("spread", "reference"):
        raise ValueError(f"shard tail '{tail}': 'spread' or 'reference'")
    chunk, rem = divmod(n_items, world)
    if tail == "reference":
        return rank * chunk, (n_items if rank == world - 1 else (rank + 1) * chunk)
    lo = rank * chunk + min(rank, rem)
    return lo, lo + chunk + (1 if rank < rem else 0)


def broadcast_static(auto, src=0, device=None):
    """rank `src` passes its built StaticAutomaton, the others pass None; every rank returns an uploaded automaton.
    With a CUDA device the four image regions travel GPU-to-GPU (RCCL) into torch-owned buffers that the handle
    adopts; on CPU (gloo, tests) the host image is broadcast and rebuilt."""
    rank = dist.get_rank()
    use_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    info = torch.zeros(8, dtype=torch.int64, device=dev)
    if rank == src:
        info.copy_(torch.from_numpy(auto.info_array()))
    dist.broadcast(info, src)
    info_h = info.cpu().numpy()
    sizes = [int(info_h[0]) * 64, int(info_h[3]) * 4, int(info_h[2]) * 8, int(info_h[6]) * 4]
    import time
    regions = []
    t_bc = time.perf_counter()
    host = auto.host_image() if rank == src else None
    for i, nbytes in enumerate(sizes):
        if rank == src:
            t = torch.from_numpy(host[i]).to(dev) if nbytes else torch.zeros(0, dtype=torch.uint8, device=dev)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            dist.broadcast(t, src)
        regions.append(t)
    if use_gpu:
        torch.cuda.synchronize()
    t_adopt = time.perf_counter()
    # what a rank pays AFTER the image has arrived: the derived walk tables (chain words, hot words + edge blocks or the edge table, the
    # bigram table, top-k counts) are re-made per rank from the adopted image -- 4-14 GB of table fill at 2^24 corpus tokens -- and are
    # reported separately from the broadcast (`distribution` attribute; bench.py static_sam_distribution)
    out = StaticAutomaton.adopt_device(info_h, regions) if use_gpu else StaticAutomaton.from_host_image(info_h, [r.numpy() for r in regions])
    if use_gpu:
        torch.cuda.synchronize()
    t_end = time.perf_counter()
    out.distribution = {"broadcast_ms": round((t_adopt - t_bc) * 1e3, 2), "derive_ms": round((t_end - t_adopt) * 1e3, 2)}
    return out


def gather_results(rows, pad=-1, device=None):
    """all-gather of ragged int rows (one per local request: output ids, or [new_tokens, steps, ...] stats) ->
    list over ranks of lists of rows.  Padded to the global max length, lengths travel alongside."""
    world = dist.get_world_size()
    use_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    n_local = torch.tensor([len(rows), max([len(r) for r in rows], default=0)], dtype=torch.int64, device=dev)
    dims = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(dims, n_local)
    n_max = max(int(d[0]) for d in dims)
    l_max = max(int(d[1]) for d in dims)
    buf = torch.full((n_max, l_max + 1), pad, dtype=torch.int64, device=dev)
    for i, r in enumerate(rows):
        buf[i, 0] = len(r)
        if len(r):
            buf[i, 1:1 + len(r)] = torch.as_tensor(np.asarray(r, dtype=np.int64)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for w in range(world):
        o = out[w].cpu().numpy()
        res.append([o[i, 1:1 + o[i, 0]].tolist() for i in range(int(dims[w][0]))])
    return res


def reduce_throughput(tokens, seconds, extra=None):
    """bench.py's whole-job figure: tokens summed over ranks, wall time = MAX over ranks, plus what every rank did.
    -> (tokens_total, seconds_max, [{"rank", "tokens", "seconds", **extra}]).  `extra`: name -> float of this rank (e.g. how long
    the static automaton took to arrive), reported per rank as is.  Works without a process group (world size 1)."""
    extra = dict(extra or {})
    names = sorted(extra)
    if not (dist.is_available() and dist.is_initialized()):
        return float(tokens), float(seconds), [dict({"rank": 0, "tokens": int(tokens), "seconds": round(float(seconds), 4)},
                                                    **{k: round(float(extra[k]), 3) for k in names})]
    use_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    mine = torch.tensor([float(tokens), float(seconds)] + [float(extra[k]) for k in names], dtype=torch.float64, device=dev)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    per_rank = [dict({"rank": r, "tokens": int(v[0].item()), "seconds": round(float(v[1].item()), 4)},
                     **{k: round(float(v[2 + i].item()), 3) for i, k in enumerate(names)}) for r, v in enumerate(parts)]
    return float(sum(v[0].item() for v in parts)), float(max(v[1].item() for v in parts)), per_rank
