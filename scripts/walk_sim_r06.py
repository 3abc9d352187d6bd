"""walk_sim_r06.py -- which REQUESTS of the shipped round-5 walk rule (edge table) could a per-CU LDS table of hot states answer?
CPU replay of the batched-walk workload (bench corpus, 10 % noise) through the rule, request by request, each tagged with the state it
addresses; then, for a budget of N staged states chosen by a STATIC score the upload can compute (occurrence count cnt_endpos of the
state, i.e. how often the corpus is in that state), the fraction of requests that fall on a staged state.  A staged state holds its
header (link, length, length of the link target) and ALL its edges when they fit the per-state budget, so hits AND conclusive misses are
answered from LDS; a state with more edges than the budget stages its most frequent ones (a miss then still goes to memory).
usage: python scripts/walk_sim_r06.py [markov|zipf] [corpus_tokens] [cursors]"""
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
link, length, deg, cnt = ex["link"].tolist(), ex["length"].tolist(), ex["deg"].tolist(), ex["aux"].tolist()
et, ed = ex["edge_tok"].tolist(), ex["edge_dst"].tolist()
edges, k = [], 0
for d in deg:
    edges.append(list(zip(et[k:k + d], ed[k:k + d]))); k += d
emap = [dict(e) for e in edges]
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

# every request: (kind, state it addresses, token, hit?)   kinds: 'bigram' (state = the root child), 'edge', 'w0', 'chain'
REQ = []
visited = 0
for b in range(B):
    idx = ln = 0; cw = []; used = 0; ptok = -1; on_child = False; at_hub = False
    for t in range(T):
        tok = toks[b][t]
        if cw and cw[0][0] == tok:
            idx += 1; ln += 1; cw = cw[1:]; used += 1; visited += 1; at_hub = False
            if used == W: cw, used = chain_word(idx), 0; REQ.append(("chain", idx, -1, True))
            ptok = tok; on_child = False; continue
        climbing = bool(cw) and cw[0][1] and ptok >= 0
        cw, used = [], 0
        if climbing or on_child:
            a = ptok if climbing else on_child_tok
            if climbing: visited += 1
            child = root_child[a]
            nx = emap[child].get(tok, -1)
            REQ.append(("bigram", child, tok, nx >= 0))
            visited += 1
            if nx >= 0:
                ln = (length[child] if climbing else ln) + 1; idx = nx; cw = chain_word(idx, 4); used = W - 4; at_hub = deg[idx] > 1
                on_child = False
            else:
                visited += 1; at_hub = False
                if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                else: idx = ln = 0; on_child = False
            ptok = tok; continue
        if idx == 0:
            visited += 1; at_hub = False
            if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
            else: idx = ln = 0
            ptok = tok; continue
        visited += 1
        done = False
        if at_hub:
            nx = emap[idx].get(tok, -1)
            REQ.append(("edge", idx, tok, nx >= 0))
            if nx >= 0:
                idx = nx; ln += 1; cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1; done = True
            else:
                REQ.append(("w0", idx, -1, True))
        else:
            REQ.append(("w0", idx, -1, True))
            if e0[idx][0] == tok:
                idx = e0[idx][1]; ln += 1
                if len(chain_word(idx)) >= 2: cw, used = chain_word(idx), 0; REQ.append(("chain", idx, -1, True))
                at_hub = False
                done = True
            elif deg[idx] > 1:
                nx = emap[idx].get(tok, -1)
                REQ.append(("edge", idx, tok, nx >= 0))
                if nx >= 0: idx = nx; ln += 1; cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1; done = True
        if not done:
            idx = link[idx]
            while True:
                visited += 1
                if idx == 0:
                    at_hub = False
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                    break
                REQ.append(("w0", idx, -1, True))
                nx = emap[idx].get(tok, -1)
                REQ.append(("edge", idx, tok, nx >= 0))
                ln = length[idx]
                if nx >= 0:
                    src = idx; idx = nx; ln += 1; on_child = False
                    if deg[src] <= 1:
                        cw, used = (chain_word(idx), 0) if len(chain_word(idx)) >= 2 else ([], 0)
                        if cw: REQ.append(("chain", idx, -1, True))
                        at_hub = False
                    else:
                        cw = chain_word(idx, 2); used = W - 2; at_hub = deg[idx] > 1
                    break
                idx = link[idx]
        ptok = tok

tot = len(REQ)
print(f"{dist} {n_tok} tokens: states {n}, visited/token {visited / (B * T):.3f}, requests per visited state {tot / visited:.3f}, per stream {tot / B:.2f}")
byk = collections.Counter(r[0] for r in REQ)
for kk, v in byk.most_common():
    print(f"   {kk:8s} {v / B:6.2f} per stream  {v / tot:.3f} of requests")
# by depth (length) of the addressed state
bylen = collections.Counter(min(length[r[1]], 8) for r in REQ)
print("   requests by length of the addressed state:", {k: round(v / tot, 3) for k, v in sorted(bylen.items())})
# staging: states ranked by occurrence count; a staged state answers w0, chain(?) no -- w0 and edge/bigram requests
order = np.argsort(-np.asarray(cnt, dtype=np.int64), kind="stable")
rank_of = np.empty(n, dtype=np.int64); rank_of[order] = np.arange(n)
req_state = np.array([r[1] for r in REQ]); req_kind = np.array([r[0] for r in REQ]); req_hit = np.array([r[3] for r in REQ])
req_tok = np.array([r[2] for r in REQ])
degs = np.asarray(deg)
for EB in (4, 8, 16, 10 ** 9):                      # edges staged per state (budget); all edges if deg <= EB
    # rank of each edge inside its state by count of the destination (the builder's top-k order = the export's edge order)
    print(f"  per-state edge budget {EB if EB < 10**9 else 'all'}:")
    for N in (256, 1024, 4096, 16384, 65536):
        staged = rank_of[req_state] < N
        # bytes: header 16 B + 8 B per staged edge
        st_states = order[:N]
        n_edges = np.minimum(degs[st_states], EB).sum()
        kb = (16 * N + 8 * n_edges) / 1024
        answered = 0
        for kind in ("w0", "edge", "bigram"):
            m = staged & (req_kind == kind)
            if kind == "w0":
                answered += m.sum()
            else:
                # answered when the state's edges are complete in LDS (deg <= EB), or on a hit among the staged top-EB edges
                comp = degs[req_state] <= EB
                a = m & comp
                rest = np.nonzero(m & ~comp & req_hit)[0]
                extra = 0
                for i in rest[:200000]:
                    s, tk = int(req_state[i]), int(req_tok[i])
                    # position of tk in the state's edge order
                    pos = next((j for j, (t2, _) in enumerate(edges[s][:EB]) if t2 == tk), -1)
                    if pos >= 0: extra += 1
                answered += a.sum() + extra
        print(f"     N = {N:6d} states ({kb:8.0f} KB): {answered / tot:.3f} of the requests answered from LDS -> requests per visited {(tot - answered) / visited:.3f}")

# ---------------------------------------------------------------------------------------------------------------------
# The round-6 rule: per-state EDGE BLOCKS whose every slot (used or empty) carries the owning state's fail header {kind of link, ref of
# link (block reference / state index / token of a root child), length of link}, and a 16-byte HOT WORD per state
# H[s] = {ref(link), len(link) | flags, e0.tok or blockref(s), e0.dst}.  Every hop of a climb is ONE request:
#   link is a hub        -> one probe of its block (hit: dst; miss: the block's header names the next hop)
#   link is single       -> H[link] (its only edge + its own fail header)
#   link is a root child -> one bigram probe (conclusive)
#   link is the root     -> nothing
# and a miss at a known hub needs no node word (the header rode in the probed slot).  Blocks are laid out hottest-first (occurrence
# count per byte); a launch copies the first LDS_KB of them into LDS and answers probes below that mark there.
def run_blocks(lds_kb_list=(0, 32, 64, 128), slots_per_edge=4):
    def nslots(d):
        m = 4
        while m < slots_per_edge * d: m <<= 1
        return m
    hubs = [s for s in range(1, n) if deg[s] > 1 and link[s] != 0 or (deg[s] > 1 and link[s] == 0 and False)]
    # root children (link == 0, length 1) stay in the bigram table; every other branching state gets a block
    hubs = [s for s in range(1, n) if deg[s] > 1 and length[s] > 1]
    size = {s: nslots(deg[s]) * 16 for s in hubs}
    hot_order = sorted(hubs, key=lambda s: -cnt[s] / size[s])
    cum, pos = 0, {}
    for s in hot_order:
        pos[s] = cum; cum += size[s]
    total_block_bytes = cum
    K = collections.Counter(); visited2 = 0
    lds_hits = {kb: 0 for kb in lds_kb_list}
    def probe(s):
        K["block probe"] += 1
        for kb in lds_kb_list:
            if pos[s] + size[s] <= kb * 1024: lds_hits[kb] += 1
    def is_hub(s): return deg[s] > 1 and length[s] > 1
    for b in range(B):
        idx = ln = 0; cw = []; used = 0; on_child = False; known_hub = False
        for t in range(T):
            tok = toks[b][t]
            if cw and cw[0][0] == tok:
                idx += 1; ln += 1; cw = cw[1:]; used += 1; visited2 += 1; known_hub = False
                if used == W: cw, used = chain_word(idx), 0; K["chain word (prefetched)"] += 1
                on_child = False; continue
            flagged = bool(cw) and cw[0][1]
            cw, used = [], 0
            if on_child:
                child = root_child[on_child_tok]
                nx = emap[child].get(tok, -1)
                K["bigram probe"] += 1; visited2 += 1
                if nx >= 0:
                    ln += 1; idx = nx; cw = chain_word(idx, 4); used = W - 4; known_hub = is_hub(idx); on_child = False
                else:
                    visited2 += 1; known_hub = False
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                continue
            if idx == 0:
                visited2 += 1; known_hub = False
                if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                else: idx = ln = 0
                continue
            # --- the cursor's own state
            visited2 += 1
            cur = idx; found = False
            if flagged:
                pass                                           # chain flag: single, link is a root child -> straight to the bigram probe below
            elif known_hub:
                probe(cur)
                nx = emap[cur].get(tok, -1)
                if nx >= 0: found = True
            else:
                K["hot word H[s] (first)"] += 1
                if deg[cur] <= 1:
                    if e0[cur][0] == tok: nx = e0[cur][1]; found = True
                else:
                    probe(cur)                                 # dependent: H[s] named the block
                    nx = emap[cur].get(tok, -1)
                    if nx >= 0: found = True
            if found:
                src = cur; idx = nx; ln += 1; on_child = False
                if deg[src] <= 1:
                    cw, used = (chain_word(idx), 0) if len(chain_word(idx)) >= 2 else ([], 0)
                    if cw: K["chain word (after e0)"] += 1
                    known_hub = False
                else:
                    cw = chain_word(idx, 2); used = W - 2; known_hub = is_hub(idx)
                continue
            # --- climb: every hop one request
            p = link[cur]
            while True:
                visited2 += 1
                if p == 0:
                    known_hub = False
                    if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                    else: idx = ln = 0; on_child = False
                    break
                if link[p] == 0 and length[p] == 1:            # a root child: one bigram probe, conclusive
                    K["bigram probe (hop)"] += 1
                    nx = emap[p].get(tok, -1)
                    if nx >= 0:
                        ln = length[p] + 1; idx = nx; cw = chain_word(idx, 4); used = W - 4; known_hub = is_hub(idx); on_child = False
                    else:
                        visited2 += 1; known_hub = False
                        if tok in root_child: idx = root_child[tok]; ln = 1; on_child = True; on_child_tok = tok
                        else: idx = ln = 0; on_child = False
                    break
                if is_hub(p): probe(p)
                else: K["hot word H[link] (hop)"] += 1
                nx = emap[p].get(tok, -1)
                if nx >= 0:
                    ln = length[p] + 1; idx = nx; on_child = False
                    if deg[p] <= 1:
                        cw, used = (chain_word(idx), 0) if len(chain_word(idx)) >= 2 else ([], 0)
                        if cw: K["chain word (after e0)"] += 1
                        known_hub = False
                    else:
                        cw = chain_word(idx, 2); used = W - 2; known_hub = is_hub(idx)
                    break
                p = link[p]
    tot2 = sum(K.values())
    assert visited2 == visited, (visited2, visited)
    print(f"--- round-6 rule (blocks with fail headers, {slots_per_edge} slots per edge: {total_block_bytes / 2**20:.0f} MB of blocks for {len(hubs)} hubs): "
          f"requests per visited {tot2 / visited:.3f} (shipped rule {tot / visited:.3f}), per stream {tot2 / B:.2f}")
    for kk, v in sorted(K.items(), key=lambda x: -x[1]):
        print(f"    {kk:32s} {v / B:7.3f} per stream")
    for kb in lds_kb_list:
        if kb: print(f"    hottest {kb:4d} KB of blocks in LDS: {lds_hits[kb] / B:6.3f} probes per stream answered there -> requests per visited {(tot2 - lds_hits[kb]) / visited:.3f}")
run_blocks()

# ---- an LDS cache of the N hottest EDGES (bigram entries and block slots alike), chosen by the static count of the edge's target ---------
def edge_cache():
    probes = [(r[1], r[2], r[3]) for r in REQ if r[0] in ("bigram", "edge")]
    allp = len(probes)
    hits = [(s, tk) for s, tk, h in probes if h]
    print(f"--- edge probes (bigram + block) {allp / B:.2f} per stream, of which hits {len(hits) / B:.2f}")
    # static ranking of every edge of every branching state by cnt[dst]
    cand = []
    for s in range(1, n):
        if deg[s] > 1:
            for tk, d in edges[s]:
                cand.append((cnt[d], s, tk))
    cand.sort(reverse=True)
    for N in (1024, 2048, 4096, 8192, 65536):
        top = set((s, tk) for _, s, tk in cand[:N])
        got = sum(1 for k in hits if k in top)
        print(f"    {N:6d} hottest edges in LDS ({N * 16 // 1024} KB at 16 B, {N * 32 // 1024} KB at load 1/2): {got / B:.3f} probes per stream answered = {got / allp:.3f} of the edge probes")
edge_cache()
