"""CPU study (no GPU): how often does the batched walk of bench.py step from state s to its rank-0 successor, and how
often is that successor the NEXT state in memory -- under the builder's creation order and under a heavy-path order.
Used to decide whether a chain-contiguous node order is worth building (walk kernel)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sam-decoding_amd")); sys.path.insert(0, ROOT)
import samd_hip, bench

n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
T = 16
flat, off, docs = bench.synth_corpus(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0) if hasattr(samd_hip.StaticAutomaton, "build_flat") else \
    samd_hip.StaticAutomaton.build([flat[off[i]:off[i + 1]] for i in range(len(off) - 1)], bench.EOS, 0)
ex = sam.export()
n = len(ex["link"])
nodes = sam.host_image()[0].view(np.int32).reshape(n, 16)
e0_dst = nodes[:, 3].copy(); e0_tok = nodes[:, 2].copy()
link, length = ex["link"], ex["length"]
deg = ex["deg"]
start = np.zeros(n + 1, np.int64); start[1:] = np.cumsum(deg)
src = np.repeat(np.arange(n, dtype=np.int64), deg)
key = src * bench.VOCAB + ex["edge_tok"]
order = np.argsort(key); key_s = key[order]; dst_s = ex["edge_dst"][order]

def step(state, tok):
    k = state.astype(np.int64) * bench.VOCAB + tok
    p = np.searchsorted(key_s, k); p = np.minimum(p, len(key_s) - 1)
    hit = key_s[p] == k
    return np.where(hit, dst_s[p], -1)

rng = np.random.default_rng(7)
n_docs, doc_len = docs.shape
d = rng.integers(0, n_docs, B); s0 = rng.integers(0, doc_len - T, B)
toks = docs[d[None, :], (s0[None, :] + np.arange(T)[:, None])]
noise = rng.random((T, B)) < 0.10
toks = np.where(noise, rng.integers(3, bench.VOCAB, (T, B)), toks)

def simulate(pos):
    """pos[s] = memory slot of state s.  Returns visits, same-line-hit counts for 16-B hot words in 64-B / 128-B groups."""
    state = np.zeros(B, np.int64)
    visits = 0; root_visits = 0
    hits = {4: 0, 8: 0}; rank0 = 0; adj = 0
    cached = {4: np.full(B, -1, np.int64), 8: np.full(B, -1, np.int64)}
    for t in range(T):
        tok = toks[t]
        todo = np.ones(B, bool)
        while todo.any():
            idx = np.nonzero(todo)[0]
            st = state[idx]
            nonroot = st != 0
            visits += int(nonroot.sum()); root_visits += int((~nonroot).sum())
            for g in (4, 8):
                line = pos[st] // g
                h = (line == cached[g][idx]) & nonroot
                hits[g] += int(h.sum())
                cached[g][idx] = np.where(nonroot, line, cached[g][idx])
            nxt = step(st, tok[idx])
            ok = nxt >= 0
            r0 = ok & (e0_tok[st] == tok[idx]) & nonroot
            rank0 += int(r0.sum()); adj += int((r0 & (pos[np.maximum(nxt, 0)] == pos[st] + 1)).sum())
            # fall back along suffix links on a miss; root miss stays at root
            fall = ~ok & (st != 0)
            state[idx] = np.where(ok, nxt, np.where(fall, link[st], 0))
            todo[idx] = fall
    return visits, root_visits, rank0, adj, hits

def heavy_path_order():
    """greedy chain layout: walk states in creation order; an unplaced state starts a chain that follows rank-0
    successors while they are unplaced."""
    pos = np.full(n, -1, np.int64); nxt_slot = 0
    e0 = e0_dst
    for s in range(n):
        while s >= 0 and pos[s] < 0:
            pos[s] = nxt_slot; nxt_slot += 1
            s = e0[s] if e0_tok[s] >= 0 else -1
    return pos

for name, pos in (("creation", np.arange(n, dtype=np.int64)), ("heavy-path", heavy_path_order())):
    v, rv, r0, adj, hits = simulate(pos)
    print(f"{name:11s} states {n} non-root visits {v} root {rv} rank0-transitions {r0} ({r0 / v:.2f}/visit) adjacent {adj} ({adj / v:.2f}/visit) "
          f"line-hit 64B {hits[4] / v:.3f} 128B {hits[8] / v:.3f} -> requests/visit {1 - hits[4] / v:.3f} / {1 - hits[8] / v:.3f}")
