"""walk_sim_r04.py -- CPU replay of bench.py's batched-walk workload through round 4's transition rule (flagged chain entries,
root-child hash with half words): HBM requests by kind per visited state and per stream, dependent rounds per wave.  See
profiles/r04_walk.md.  (It models the per-child hashed blocks of the first half of round 4; with the bigram table that replaced them the
"root(L2)" lines below cost no request at all and probes of small root children are table look-ups too.)   usage: python scripts/walk_sim_r04.py [corpus_tokens] [cursors] [half-word entries]"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sam-decoding_amd")); sys.path.insert(0, ROOT)
import samd_hip, bench
n_tok = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
E = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T=16; W=8
flat, off, docs = bench.synth_corpus(n_tok)
sam = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, 0)
ex = sam.export()
n = len(ex["link"])
link, length, deg = ex["link"].tolist(), ex["length"].tolist(), ex["deg"].tolist()
et, ed = ex["edge_tok"].tolist(), ex["edge_dst"].tolist()
edges, k = [], 0
for d in deg:
    edges.append(list(zip(et[k:k + d], ed[k:k + d]))); k += d
emap = [dict(e) for e in edges]
e0 = [(e[0] if e else (-1, -1)) for e in edges]
is_chain = [e0[s][1] == s + 1 and e0[s][0] >= 0 for s in range(n)]
flag = [deg[s] <= 1 and link[s] > 0 and link[link[s]] == 0 for s in range(n)]
print("flagged states", sum(flag)/n)
def chain_word(s, w=W):
    out = []
    while len(out) < w and s < n and is_chain[s]:
        out.append(e0[s][0]); s += 1
    return out
rng = np.random.default_rng(7)
n_docs, doc_len = docs.shape
d = rng.integers(0, n_docs, B); s0 = rng.integers(0, doc_len - T, B)
toks = docs[d[None, :], (s0[None, :] + np.arange(T)[:, None])]
noise = rng.random((T, B)) < 0.10
toks = np.where(noise, rng.integers(3, bench.VOCAB, (T, B)), toks).T.tolist()
K = collections.Counter(); visited = 0; rounds=np.zeros((B,T),np.int32)
for b in range(B):
    idx = ln = 0; cw=None; used=0; rc=False; ptok=-1
    for t in range(T):
        tok = toks[b][t]; r=0
        if cw and cw[0]==tok:
            idx+=1; ln+=1; cw=cw[1:]; used+=1; visited+=1; rc=False
            if used==W: cw,used=chain_word(idx),0; K["chain(full word)"]+=1; r+=1
            ptok=tok; rounds[b][t]=r; continue
        have = cw is not None; had_tok = bool(cw)
        cw,used=None,0
        hopped=False
        if rc:
            K["d1probe"]+=1; r+=1
            rc=False
            nx = emap[idx].get(tok,-1)
            if nx>=0:
                idx=nx; ln+=1; visited+=1; cw=chain_word(idx,E); used=W-E
                ptok=tok; rounds[b][t]=r; continue
            visited+=1; idx=0; ln=0; hopped=True
        elif have and had_tok and flag[idx] and ptok>=0:
            # flagged climb: root16[ptok] (L2) -> probe
            visited+=1  # idx itself
            l=link[idx]; assert l==emap[0][ptok]
            K["root(L2)"]+=1; r+=1
            if deg[l]>5:
                K["d1probe(climb)"]+=1; r+=1
                nx=emap[l].get(tok,-1)
                visited+=1
                if nx>=0:
                    idx=nx; ln=length[l]+1; cw=chain_word(idx,E); used=W-E; ptok=tok; rounds[b][t]=r; continue
                idx=0; ln=0; hopped=True
            else:
                idx=l; ln=length[l]; hopped=True
        while True:
            visited+=1
            if idx==0:
                K["root(L2)"]+=1; r+=1
                nx=emap[0].get(tok,-1)
                if nx>=0: idx,ln=nx,ln+1; rc = deg[nx]>5
                else: idx,ln=0,0
                break
            K["node"+("(climb)" if hopped else "")]+=1; r+=1
            if hopped: ln=length[idx]
            if e0[idx][0]==tok:
                idx=e0[idx][1]; ln+=1
                if len(chain_word(idx))>=2: cw,used=chain_word(idx),0; K["chain(after e0)"]+=1; r+=1
                break
            nx=-1
            if deg[idx]>1:
                K["tail"]+=1; r+=1
                nx=emap[idx].get(tok,-1)
                if deg[idx]>5 and (nx<0 or [e[0] for e in edges[idx]].index(tok)>=5): K["spill"]+=1; r+=1
            if nx>=0:
                idx=nx; ln+=1; cw,used=chain_word(idx),0; K["chain(after e>=1)"]+=1; r+=1
                break
            idx=link[idx]; hopped=True
            if idx==0: ln=0
        ptok=tok; rounds[b][t]=r
print("states",n,"visited/token",visited/(B*T))
tot=sum(v for k,v in K.items() if "L2" not in k)
for k,v in sorted(K.items(), key=lambda x:-x[1]): print(f"{k:22s} {v/visited:.4f} per visit   {v/(B):.3f} per stream")
print("HBM loads/visit (tails 0.5):", (tot - 0.5*K['tail'])/visited)
waves = rounds.reshape(B // 64, 64, T)
print("rounds per wave lock-step", waves.max(axis=1).sum(axis=1).mean(), "mean lane", waves.sum(axis=2).mean())
