"""CPU study (no GPU): memory requests per visited state of the batched walk under a "chain word" layout.

Today every visited state costs one 16-byte load (its hot word w0) = one memory request, and the kernel runs at the memory
system's request rate (profiles/r01_walk_pmc.md, r01_hbm_probe.md: a second load from the same 128-byte block does not come
cheaper, so fetching a neighbour speculatively does not pay).  What CAN remove requests is carrying more future transitions in
the 16 bytes a lane already loads: for a state s on a non-branching run (rank-0 successor of s + i is s + i + 1 -- 61 % of this
workload's transitions), `chain[s]` = the next 8 rank-0 tokens as u16.  A lane that holds chain[s] follows up to 8 matching
tokens with no memory access; only a mismatch (or the end of the word) goes back to the 64-byte node.

This script replays bench.py's walk workload on the host image and counts requests per visited state for:
  now      one w0 load per visited state
  chain8   load chain[s'] when a transition lands on a state flagged "run of >= RUN_MIN rank-0 steps ahead" (flag bit in the edge
           target), else the node; a mismatch inside the word costs the node load of that state
usage: python scripts/walk_chain_sim.py [corpus_tokens] [cursors] [run_min]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sam-decoding_amd")); sys.path.insert(0, ROOT)
import samd_hip, bench

n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
RUN_MIN = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T, W = 16, 8
flat, off, docs = bench.synth_corpus(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0)
ex = sam.export()
n = len(ex["link"])
nodes = sam.host_image()[0].view(np.int32).reshape(n, 16)
e0_tok, e0_dst = nodes[:, 2].copy(), nodes[:, 3].copy()
link, deg = ex["link"], ex["deg"]
src = np.repeat(np.arange(n, dtype=np.int64), deg)
key = src * bench.VOCAB + ex["edge_tok"]
order = np.argsort(key); key_s = key[order]; dst_s = ex["edge_dst"][order]
is_chain = (e0_dst == np.arange(n) + 1) & (e0_tok >= 0)            # rank-0 successor is the next state in memory
# run[s] = number of consecutive chain steps starting at s (capped at W)
run = np.zeros(n + 1, np.int64)
for s in range(n - 1, -1, -1):
    run[s] = min(W, run[s + 1] + 1) if is_chain[s] else 0
run = run[:n]
print(f"states {n}: chain states {is_chain.mean():.3f}, run >= 2: {(run >= 2).mean():.3f}, run >= 4: {(run >= 4).mean():.3f}, run == 8: {(run >= 8).mean():.3f}")

def step(state, tok):
    k = state.astype(np.int64) * bench.VOCAB + tok
    p = np.minimum(np.searchsorted(key_s, k), len(key_s) - 1)
    return np.where(key_s[p] == k, dst_s[p], -1)

rng = np.random.default_rng(7)
n_docs, doc_len = docs.shape
d = rng.integers(0, n_docs, B); s0 = rng.integers(0, doc_len - T, B)
toks = docs[d[None, :], (s0[None, :] + np.arange(T)[:, None])]
toks = np.where(rng.random((T, B)) < 0.10, rng.integers(3, bench.VOCAB, (T, B)), toks)

state = np.zeros(B, np.int64)
left = np.zeros(B, np.int64)             # chain tokens still held in registers for the cursor's current state (0 = none)
visits = req_now = req_chain = chain_loads = free_hits = root = 0
for t in range(T):
    tok = toks[t]
    todo = np.ones(B, bool)
    first = np.ones(B, bool)             # first visit of this token (link hops afterwards always need the node)
    while todo.any():
        idx = np.nonzero(todo)[0]
        st = state[idx]
        nonroot = st != 0
        visits += int(nonroot.sum()); root += int((~nonroot).sum())
        req_now += int(nonroot.sum())
        nxt = step(st, tok[idx])
        hit0 = nonroot & (nxt >= 0) & (e0_tok[st] == tok[idx]) & is_chain[st]     # the transition the chain word encodes
        have = (left[idx] > 0) & first[idx] & nonroot
        free = have & hit0
        free_hits += int(free.sum())
        # everything else at a non-root state loads the node's w0 (a held chain word that mismatches included)
        req_chain += int((nonroot & ~free).sum())
        ok = nxt >= 0
        fall = ~ok & nonroot
        new_state = np.where(ok, nxt, np.where(fall, link[st], 0))
        # registers after the move: a free hit consumes one chain token; any other landing reloads per the flag rule
        new_left = np.where(free, left[idx] - 1, 0)
        landed = ok & ~free                                        # followed an edge out of a node (or the root table)
        want = landed & (run[np.maximum(new_state, 0)] >= RUN_MIN) & (new_state != 0)
        chain_loads += int(want.sum())
        new_left = np.where(want, run[np.maximum(new_state, 0)], new_left)
        # a word that ran out while the run continues: reload at the next state if it is still flagged
        ran_out = free & (new_left == 0) & (run[np.maximum(new_state, 0)] >= RUN_MIN)
        chain_loads += int(ran_out.sum())
        new_left = np.where(ran_out, run[np.maximum(new_state, 0)], new_left)
        state[idx] = new_state
        left[idx] = np.where(fall, 0, new_left)
        first[idx] = False
        todo[idx] = fall
req_chain += chain_loads
print(f"visited non-root states {visits} (root {root}); requests now {req_now} = {req_now / visits:.3f}/visit; "
      f"chain8(run_min {RUN_MIN}): {req_chain} = {req_chain / visits:.3f}/visit (chain-word loads {chain_loads}, register hits {free_hits} = {free_hits / visits:.3f}/visit)")
