"""walk_sched_sim.py -- CPU replay of bench.py's batched-walk workload (B cursors x 16 tokens from the root) through the exact
transition rule of k_static_walk (st_transfer_chain, csrc/sam_device.h), counting per lane and token the DEPENDENT memory rounds
(one round = loads issued, then waited for) and the loads by kind.  Two questions:

  1. what does a wave cost in round trips when its 64 lanes advance token by token in lock-step (today: the wave waits at every
     token until its slowest lane has resolved it) against lanes that run decoupled (a wave is done when its slowest LANE is)?
  2. how many loads do variants of the data layout remove -- `jump` links that skip suffix-link ancestors with the same edge set,
     chain words of 16 tokens, a bigram table for the first two tokens after the root?

usage: python scripts/walk_sched_sim.py [corpus_tokens] [cursors]       (pure host; ~1 min at 2^20 / 8192)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sam-decoding_amd")); sys.path.insert(0, ROOT)
import samd_hip, bench

n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = 16
flat, off, docs = bench.synth_corpus(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0)
ex = sam.export()
n = len(ex["link"])
link, length, deg = ex["link"].tolist(), ex["length"].tolist(), ex["deg"].tolist()
et, ed = ex["edge_tok"].tolist(), ex["edge_dst"].tolist()
edges, k = [], 0
for d in deg:
    edges.append(list(zip(et[k:k + d], ed[k:k + d]))); k += d          # top-k order first (rank 0 = most frequent)
emap = [dict(e) for e in edges]
e0 = [(e[0] if e else (-1, -1)) for e in edges]
is_chain = [e0[s][1] == s + 1 and e0[s][0] >= 0 for s in range(n)]


def chain_word(s, W):
    out = []
    while len(out) < W and s < n and is_chain[s]:
        out.append(e0[s][0]); s += 1
    return out


# jump[s]: first suffix-link ancestor whose edge set is LARGER than s's (edge sets are nested along suffix links, so an ancestor of
# the same degree has the same edges: a token that has no edge at s has none there either); hops[s] = link hops it stands for
jump, hops = [0] * n, [1] * n
order = sorted(range(1, n), key=lambda s: length[s])
for s in order:
    p = link[s]
    if p > 0 and deg[p] == deg[s]:
        jump[s], hops[s] = jump[p], hops[p] + 1
    else:
        jump[s], hops[s] = p, 1

rng = np.random.default_rng(7)
n_docs, doc_len = docs.shape
d = rng.integers(0, n_docs, B); s0 = rng.integers(0, doc_len - T, B)
toks = docs[d[None, :], (s0[None, :] + np.arange(T)[:, None])]
toks = np.where(rng.random((T, B)) < 0.10, rng.integers(3, bench.VOCAB, (T, B)), toks).T.tolist()


def simulate(W=8, use_jump=False, bigram=False):
    """-> rounds[b][t] (dependent memory rounds of lane b for token t), loads by kind, visited states (the reference's count)"""
    rounds = np.zeros((B, T), np.int32)
    kinds = dict(root=0, node=0, tail=0, spill=0, chain=0, bigram=0)
    visited = 0
    for b in range(B):
        idx = ln = 0
        cw, used = [], 0
        pending_chain = 0                                  # a chain-word load issued after the previous token: waited for at this one
        row = toks[b]
        for t in range(T):
            tok = row[t]
            r = pending_chain; pending_chain = 0
            if cw and cw[0] == tok:                         # register path
                idx += 1; ln += 1; cw = cw[1:]; used += 1; visited += 1
                if used == W:
                    cw, used = chain_word(idx, W), 0; kinds["chain"] += 1; pending_chain = 1
                rounds[b][t] = r
                continue
            cw, used = [], 0
            hopped = False
            while True:
                visited += 1
                if idx == 0:
                    if bigram and t + 1 < T and not hopped and False:
                        pass
                    kinds["root"] += 1; r += 1
                    nx = emap[0].get(tok, -1)
                    if nx >= 0: idx, ln = nx, ln + 1
                    else: idx, ln = 0, 0
                    break
                kinds["node"] += 1; r += 1
                if hopped: ln = length[idx]
                if e0[idx][0] == tok:
                    src = idx; idx = e0[idx][1]; ln += 1
                    if len(chain_word(idx, W)) >= 2:        # SAMD_RUN
                        cw, used = chain_word(idx, W), 0; kinds["chain"] += 1; pending_chain = 1
                    break
                nx = -1
                if deg[idx] > 1:
                    kinds["tail"] += 1; r += 1             # w1..w3: same 64-byte line, a second dependent round
                    nx = emap[idx].get(tok, -1)
                    if deg[idx] > 5 and (nx < 0 or [e[0] for e in edges[idx]].index(tok) >= 5):
                        kinds["spill"] += 1; r += 1        # ~1.5 probes of a hashed block: count one round, one line
                if nx >= 0:
                    idx = nx; ln += 1
                    cw, used = chain_word(idx, W), 0; kinds["chain"] += 1; pending_chain = 1
                    break
                if use_jump:
                    visited += hops[idx] - 1; idx = jump[idx]
                else:
                    idx = link[idx]
                hopped = True
                if idx == 0: ln = 0
            rounds[b][t] = r
    return rounds, kinds, visited


def report(name, rounds, kinds, visited):
    waves = rounds.reshape(B // 64, 64, T)
    lock = waves.max(axis=1).sum(axis=1)                   # per wave: sum over tokens of the slowest lane's rounds
    free = waves.sum(axis=2).max(axis=1)                   # per wave: the slowest lane's total
    loads = sum(kinds.values())
    hbm = kinds["node"] + kinds["spill"] + kinds["chain"]  # distinct 64-byte lines (tails share their node's line, the root table is L2-resident)
    print(f"{name:34s} visited {visited / (B * T):.3f}/token  loads {loads / visited:.3f}/visit  lines {hbm / visited:.3f}/visit "
          f"[node {kinds['node'] / visited:.3f} tail {kinds['tail'] / visited:.3f} spill {kinds['spill'] / visited:.3f} chain {kinds['chain'] / visited:.3f} root {kinds['root'] / visited:.3f}]  "
          f"rounds per wave: lock-step {lock.mean():.1f}, decoupled {free.mean():.1f} (mean lane {waves.sum(axis=2).mean():.1f})")


print(f"corpus {n_tok} tokens, {n} states; {B} cursors x {T} tokens")
for name, kw in (("today (8-token chain words)", dict()), ("+ jump links", dict(use_jump=True)), ("16-token chain words", dict(W=16)),
                 ("16-token words + jump links", dict(W=16, use_jump=True)), ("no chain words", dict(W=0))):
    if kw.get("W", 8) == 0:
        save = is_chain; is_chain = [False] * n
        report(name, *simulate(W=8)); is_chain = save
    else:
        report(name, *simulate(**kw))
