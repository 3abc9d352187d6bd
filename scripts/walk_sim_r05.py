"""walk_sim_r05.py -- CPU replay of the batched-walk workload through the round-4 transition rule (st_transfer_chain: chain words,
flagged entries, bigram table) and through round 5's EDGE TABLE rule on the Markov AND the Zipf corpus: dependent memory ROUNDS per
(lane, token), what a lock-step wave of 64 lanes pays (the max over its lanes, per token), and requests by kind; then priced variants of the
shipped rule (run_edge's docstring).  profiles/r05_walk_sweep.md quotes its numbers.
usage: python scripts/walk_sim_r05.py [markov|zipf] [corpus_tokens] [cursors]"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sam-decoding_amd")); sys.path.insert(0, ROOT)
import samd_hip, bench
dist = sys.argv[1] if len(sys.argv) > 1 else "zipf"
n_tok = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T = 16; W = 8
flat, off, docs = bench.synth_corpus(n_tok) if dist == "markov" else bench.synth_corpus_zipf(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0)
ex = sam.export()
n = len(ex["link"])
link, length, deg = ex["link"].tolist(), ex["length"].tolist(), ex["deg"].tolist()
et, ed = ex["edge_tok"].tolist(), ex["edge_dst"].tolist()
edges, k = [], 0
for d in deg:
    edges.append(list(zip(et[k:k + d], ed[k:k + d]))); k += d
emap = [dict(e) for e in edges]
rank = [{t: r for r, (t, _) in enumerate(e)} for e in edges]
e0 = [(e[0] if e else (-1, -1)) for e in edges]
is_chain = [e0[s][1] == s + 1 and e0[s][0] >= 0 for s in range(n)]
flag = [deg[s] <= 1 and link[s] > 0 and link[link[s]] == 0 for s in range(n)]
def chain_word(s, w=W):
    out = []
    while len(out) < w and s < n and is_chain[s]:
        out.append((e0[s][0], flag[s])); s += 1
    return out
rng = np.random.default_rng(7)
n_docs, doc_len = docs.shape
dd = rng.integers(0, n_docs, B); s0 = rng.integers(0, doc_len - T, B)
toks = docs[dd[None, :], (s0[None, :] + np.arange(T)[:, None])]
noise = rng.random((T, B)) < 0.10
if dist == "markov":
    ntok = rng.integers(3, bench.VOCAB, (T, B))
else:
    ntok = (3 + np.searchsorted(bench.zipf_cdf(bench.VOCAB), rng.random((T, B)), side="right")).clip(3, bench.VOCAB - 1)
toks = np.where(noise, ntok, toks).T.tolist()
root_child = emap[0]

def run(rule):
    """rule: 'r04' = the shipped kernel; 'hub' = every transition out of a branching state of depth >= 2 is ONE probe of a global
    (state, token) hash whose entry carries the half chain word, a miss needs the node's word 0 for its link (requested WITH the probe);
    'hub+lazy' = that word only after a miss (a dependent round)."""
    K = collections.Counter(); visited = 0
    rounds = np.zeros((B, T), np.int32)
    for b in range(B):
        idx = ln = 0; cw = []; used = 0; ptok = -1; on_child = False
        for t in range(T):
            tok = toks[b][t]; r = 0
            if cw and cw[0][0] == tok:                       # register path
                idx += 1; ln += 1; cw = cw[1:]; used += 1; visited += 1
                if used == W:
                    cw, used = chain_word(idx), 0; K["chain(next word, prefetched)"] += 1
                ptok = tok; rounds[b][t] = r; on_child = False; continue
            climbing = bool(cw) and cw[0][1] and ptok >= 0
            cw, used = [], 0
            if climbing or on_child:                         # bigram probe (one round)
                a = ptok if climbing else on_child_tok
                if climbing: visited += 1
                K["bigram probe"] += 1; r += 1
                child = root_child[a]
                nx = emap[child].get(tok, -1)
                visited += 1
                if nx >= 0:
                    ln = (length[child] if climbing else ln) + 1; idx = nx; cw = chain_word(idx, 4); used = W - 4
                    on_child = False
                else:
                    visited += 1
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                ptok = tok; rounds[b][t] = r; continue
            if idx == 0:
                visited += 1
                if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                else: idx = ln = 0
                ptok = tok; rounds[b][t] = r; continue
            hopped = False
            while True:
                visited += 1
                if idx == 0:
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                    break
                if link[idx] == 0 and hopped and rule != "r04" and False:
                    pass
                branching = deg[idx] > 1
                if rule.startswith("hub") and branching:
                    K["hub probe"] += 1; r += 1
                    if rule == "hub": K["node w0 (with the probe)"] += 1
                    nx = emap[idx].get(tok, -1)
                    if hopped: pass
                    if nx >= 0:
                        if hopped: ln = length[idx]
                        idx = nx; ln += 1; cw = chain_word(idx, 4); used = W - 4; on_child = False
                        break
                    if rule == "hub+lazy": K["node w0 (after a miss)"] += 1; r += 1
                    if hopped: ln = length[idx]
                    idx = link[idx]; hopped = True
                    if idx == 0: ln = 0
                    continue
                K["node w0"] += 1; r += 1
                if hopped: ln = length[idx]
                if e0[idx][0] == tok:
                    src = idx; idx = e0[idx][1]; ln += 1
                    if len(chain_word(idx)) >= 2: cw, used = chain_word(idx), 0; K["chain word (after e0)"] += 1; r += 1
                    on_child = False
                    break
                nx = -1
                if branching:
                    K["node w1-3 (same line)"] += 1; r += 1
                    nx = emap[idx].get(tok, -1)
                    if deg[idx] > 5 and (nx < 0 or rank[idx][tok] >= 5): K["spill probe"] += 1; r += 1
                if nx >= 0:
                    idx = nx; ln += 1; cw, used = chain_word(idx), 0; K["chain word (after e>=1)"] += 1; r += 1; on_child = False
                    break
                idx = link[idx]; hopped = True
                if idx == 0: ln = 0
            ptok = tok; rounds[b][t] = r
    waves = rounds.reshape(B // 64, 64, T)
    print(f"--- rule {rule}: visited/token {visited / (B * T):.3f}; rounds per lane-token mean {rounds.mean():.3f}; per WAVE-token (max over 64 lanes) "
          f"{waves.max(axis=1).mean():.2f}; per wave x16 tokens {waves.max(axis=1).sum(axis=1).mean():.1f}; if lanes ran decoupled {waves.sum(axis=2).max(axis=1).mean():.1f}")
    tot = sum(K.values())
    for kk, v in sorted(K.items(), key=lambda x: -x[1]):
        print(f"    {kk:32s} {v / B:7.3f} per stream   {v / visited:.4f} per visit")
    print(f"    requests per visit {tot / visited:.3f} (same-line w1-3 counted)")

def run_edge(variant):
    """the shipped round-5 rule (st_transfer_chain with the edge table), request by request.  variants:
       'shipped'      : at_hub -> probe (node only on a miss); otherwise node w0, then a probe if the state branches; every hop = w0 + probe together
       'hubterm'      : + the chain word's terminator / the e0 edge says whether the state it ends on branches (no w0 in front of the probe)
       'linkhub'      : + w0 carries 'my suffix link is a branching state': a hop to a SINGLE state skips the probe
       'fail2'        : + a hop's w0 is replaced by a per-state fail record {link, len(link), link2, len(link2)} (16 B) fetched once per TWO hops"""
    K = collections.Counter(); visited = 0
    rounds = np.zeros((B, T), np.int32)
    hubterm = variant in ("hubterm", "linkhub", "fail2")
    linkhub = variant in ("linkhub", "fail2")
    fail2 = variant == "fail2"
    for b in range(B):
        idx = ln = 0; cw = []; used = 0; ptok = -1; on_child = False; at_hub = False
        for t in range(T):
            tok = toks[b][t]; r = 0
            if cw and cw[0][0] == tok:
                idx += 1; ln += 1; cw = cw[1:]; used += 1; visited += 1; at_hub = False
                if used == W: cw, used = chain_word(idx), 0; K["chain word (prefetched)"] += 1
                if hubterm and not cw and deg[idx] > 1: at_hub = True
                ptok = tok; rounds[b][t] = r; on_child = False; continue
            climbing = bool(cw) and cw[0][1] and ptok >= 0
            cw, used = [], 0
            if climbing or on_child:
                a = ptok if climbing else on_child_tok
                if climbing: visited += 1
                K["bigram probe"] += 1; r += 1
                child = root_child[a]
                nx = emap[child].get(tok, -1)
                visited += 1
                if nx >= 0:
                    ln = (length[child] if climbing else ln) + 1; idx = nx; cw = chain_word(idx, 4); used = W - 4; at_hub = deg[idx] > 1
                    on_child = False
                else:
                    visited += 1; at_hub = False
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                ptok = tok; rounds[b][t] = r; continue
            if idx == 0:
                visited += 1; at_hub = False
                if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                else: idx = ln = 0
                ptok = tok; rounds[b][t] = r; continue
            visited += 1
            done = False
            if at_hub:
                K["edge probe (hub known)"] += 1; r += 1
                nx = emap[idx].get(tok, -1)
                if nx >= 0:
                    idx = nx; ln += 1; cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1; done = True
                else:
                    K["node w0 after a hub miss"] += 1; r += 1
            else:
                K["node w0 (first)"] += 1; r += 1
                if e0[idx][0] == tok:
                    idx = e0[idx][1]; ln += 1
                    if len(chain_word(idx)) >= 2: cw, used = chain_word(idx), 0; K["chain word (after e0)"] += 1; r += 1
                    at_hub = hubterm and deg[idx] > 1 and not cw
                    done = True
                elif deg[idx] > 1:
                    K["edge probe (after w0)"] += 1; r += 1
                    nx = emap[idx].get(tok, -1)
                    if nx >= 0: idx = nx; ln += 1; cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1; done = True
            if not done:
                hop = 0
                idx = link[idx]
                while True:
                    visited += 1
                    if idx == 0:
                        at_hub = False
                        if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                        else: idx = ln = 0; on_child = False
                        break
                    r += 1
                    if fail2:
                        if hop % 2 == 0: K["fail record (2 hops)"] += 1
                    else:
                        K["node w0 (hop)"] += 1
                    if deg[idx] > 1 or not linkhub: K["edge probe (hop)"] += 1
                    if fail2 and deg[idx] <= 1: K["node w0 (hop, single state)"] += 1
                    hop += 1
                    ln = length[idx]
                    nx = emap[idx].get(tok, -1)
                    if nx >= 0:
                        src = idx; idx = nx; ln += 1; on_child = False
                        if deg[src] <= 1:
                            cw, used = (chain_word(idx), 0) if len(chain_word(idx)) >= 2 else ([], 0)
                            if cw: K["chain word (after e0)"] += 1; r += 1
                            at_hub = hubterm and deg[idx] > 1 and not cw
                        else:
                            cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1
                        break
                    idx = link[idx]
            ptok = tok; rounds[b][t] = r
    waves = rounds.reshape(B // 64, 64, T)
    tot = sum(K.values())
    print(f"--- edge-table rule, variant {variant}: requests per visit {tot / visited:.3f}; rounds per WAVE x16 tokens {waves.max(axis=1).sum(axis=1).mean():.1f}")
    for kk, v in sorted(K.items(), key=lambda x: -x[1]):
        print(f"    {kk:32s} {v / B:7.3f} per stream   {v / visited:.4f} per visit")

print(dist, "states", n, "non-branching", sum(1 for d in deg if d <= 1) / n)
run("r04")                                                   # the round-4 rule (no edge table)
for v in ("shipped", "hubterm", "linkhub", "fail2"):
    run_edge(v)
sys.exit(0)
for rule in ("r04", "hub", "hub+lazy"):
    run(rule)
