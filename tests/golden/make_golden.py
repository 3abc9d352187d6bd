#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the Python reference.

Runs only in the dev container (needs /root/reference); the fixtures it writes are plain
data (inputs + the reference's outputs) and are what travels to the GPU box.  Nothing from
the reference's source is copied: the loader below skips the package __init__ (which needs
transformers 4.4x symbols) exactly as SURVEY.md Appendix C describes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixture files (gzip JSON):
    sam_traces.json.gz     state tables / cursors / lookups of DynSAM, StaticSAM (SO) and StaticSAM (S)
    drafts.json.gz         DynSAM.gen_draft, S-variant to_anc/gen_draft, StaticSAM.gen_draft trees, top-k tables
    buffers.json.gz        gen_buffers (SO) and token-recycle gen_buffers
    draft_model.json.gz    DraftModel.lookup/update decisions (SO and S)
    posterior.json.gz      gen_candidates tail + eval_posterior (greedy)
    token_recycle.json.gz  TokenRecycle.update / gen_draft round trips
    loop_so.json.gz        whole generate() loop traces of samd_sam_only with a scripted LM
    loop_s.json.gz         whole generate() loop traces of samd[token_recycle] with a scripted LM
"""
import gzip
import io
import json
import os
import sys
import types
import contextlib

import numpy as np
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, R)
sys.path.insert(0, os.path.dirname(HERE))
from scripted_lm import ScriptedLM as _NpScriptedLM, perm_logits
for pkg in ("samd_sam_only", "samd"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{R}/{pkg}"]
    sys.modules[pkg] = m

with contextlib.redirect_stdout(io.StringIO()):
    from samd_sam_only.sam import StaticSAM as SO_StaticSAM, DynSAM as SO_DynSAM
    from samd_sam_only.draft import DraftModel as SO_DraftModel
    from samd_sam_only.samd_config import SamdConfig as SO_SamdConfig
    from samd_sam_only.utils import gen_candidates, eval_posterior, SamdGenerationConfig
    from samd.sam import StaticSAM as S_StaticSAM, DynSAM as S_DynSAM
    from samd.draft import DraftModel as S_DraftModel
    from samd.samd_config import SamdConfig as S_SamdConfig
    from samd.tree_model.token_recycle import TokenRecycle
    from samd.tree_model.token_recycle.utils import gen_buffers as tr_gen_buffers


def quiet(fn, *a, **k):
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def dump(name, obj):
    path = os.path.join(HERE, name)
    raw = json.dumps(obj, separators=(",", ":")).encode()
    with open(path, "wb") as raw_f:
        with gzip.GzipFile(fileobj=raw_f, mode="wb", mtime=0) as f:   # reproducible bytes
            f.write(raw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- streams
def stream_zipf(rng, n, vocab=32000, a=1.2):
    return (np.minimum(rng.zipf(a, n), vocab - 3) + 2).astype(int).tolist()


def stream_markov(rng, n, vocab=200, succ=3, noise=0.05):
    """sparse order-2 Markov source: repetitive text with frequent re-occurring substrings."""
    table = {}
    out = [int(rng.integers(3, vocab)), int(rng.integers(3, vocab))]
    while len(out) < n:
        key = (out[-2], out[-1])
        if key not in table:
            table[key] = rng.integers(3, vocab, succ).tolist()
        if rng.random() < noise:
            out.append(int(rng.integers(3, vocab)))
        else:
            w = np.array([1.0 / (i + 1) for i in range(succ)])
            out.append(int(table[key][rng.choice(succ, p=w / w.sum())]))
    return out


def streams(rng):
    return {
        "zipf": stream_zipf(rng, 300),
        "markov": stream_markov(rng, 400),
        "a^n": [7] * 64,
        "(ab)^n": [5, 9] * 40,
        "distinct": list(range(10, 110)),
        "small_vocab": rng.integers(3, 7, 300).tolist(),
        "abcabx": [3, 4, 5, 3, 4, 6, 3, 4, 5, 3, 4, 5, 7, 3, 4, 6, 3, 4, 5],
    }


def state_table(sam, aux):
    return {
        "link": [s.link for s in sam.states],
        "length": [s.length for s in sam.states],
        "aux": [getattr(s, aux) for s in sam.states],
        "edges": [[[int(t), int(d)] for t, d in s.next.items()] for s in sam.states],   # dict order
    }


# --------------------------------------------------------------------------- 1-2: SAM traces
def gen_sam_traces(rng):
    out = {"dyn": [], "static_so": [], "static_s": []}
    for name, toks in streams(rng).items():
        # DynSAM: prefill chunk then small commits (as generate() does), cursor after each
        sam = SO_DynSAM(device="cpu")
        cuts = sorted(set([0, len(toks) // 2] + list(range(len(toks) // 2, len(toks), 3)) + [len(toks)]))
        cursors = []
        probes = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            sam.add_tokens(toks[a:b])
            cursors.append([sam.cur_index, sam.cur_length])
            ptoks = [toks[int(rng.integers(0, b))] for _ in range(3)] + [1]
            probes.append([[t, *sam.lookup(t)] for t in ptoks])
        out["dyn"].append({"name": name, "tokens": toks, "cuts": cuts, "cursors": cursors, "probes": probes,
                           "input_ids": sam.input_ids, "last": sam.last, "max_length": sam.max_length,
                           "table": state_table(sam, "min_endpos")})
    # static corpora: documents + optional vocab docs; queries walk the cursor
    corpora = {
        "markov_docs": ([stream_markov(rng, 60) for _ in range(12)], 2),
        "small_vocab_docs": ([rng.integers(3, 7, 40).tolist() for _ in range(10)], 2),
        "eos_terminated": ([[5, 6, 7, 2], [6, 7, 2], [6, 7, 8], [2], [1]], 2),
        "vocab_docs": ([stream_markov(rng, 50, vocab=40) for _ in range(6)] + [[i] for i in range(40)], 2),
        "ties": ([[3, 4], [3, 5], [3, 6], [3, 4], [3, 5], [3, 6], [3, 7, 8], [3, 7, 9]], 2),
    }
    for name, (docs, eos) in corpora.items():
        so = quiet(SO_StaticSAM.build, docs, eos, False)
        s = S_StaticSAM.build(docs, eos, False)
        flat = [t for d in docs for t in d]
        q = [flat[int(rng.integers(0, len(flat)))] for _ in range(10)]
        start = int(rng.integers(0, max(1, len(flat) - 30)))
        q += flat[start:start + 25] + [1, 1] + flat[5:15]
        walk_so, walk_s = [], []
        so.reset(); s.reset()
        for t in q:
            pi = so.lookup(t)
            so.transfer_tokens([t])
            assert (so.cur_index, so.cur_length) == tuple(pi)
            walk_so.append([so.cur_index, so.cur_length])
            s.transfer_tokens([t])
            walk_s.append([s.cur_index, s.cur_length])
        out["static_so"].append({"name": name, "docs": docs, "eos": eos, "query": q, "walk": walk_so,
                                 "last": so.last, "max_length": so.max_length,
                                 "table": state_table(so, "cnt_endpos"),
                                 "topk": [[[int(t), int(d)] for t, d in x] for x in so.states_topk_next]})
        out["static_s"].append({"name": name, "docs": docs, "eos": eos, "query": q, "walk": walk_s,
                                "input_ids": s.input_ids, "table": state_table(s, "min_endpos")})
    dump("sam_traces.json.gz", out)
    return corpora


# --------------------------------------------------------------------------- 3-5: drafts
def gen_drafts(rng, corpora):
    out = {"dyn_so": [], "dyn_s": [], "static_s": [], "tree": []}
    for name, toks in streams(rng).items():
        for (mp, alpha) in [(60, 4.0), (16, 4.0), (8, 0.5), (40, 1.7)]:
            sam = SO_DynSAM(max_predicts=mp, alpha=alpha, device="cpu")
            sam.add_tokens(toks)
            cases = []
            for _ in range(6):
                t = toks[int(rng.integers(0, len(toks)))]
                idx, ln = sam.lookup(t)
                seq, buf = sam.gen_draft(idx, ln, t)
                cases.append({"start": t, "index": idx, "match": ln, "seq": seq,
                              "pos": buf["seq_position_ids"].tolist()})
            # index 0 / match 0
            seq, _ = sam.gen_draft(0, 0, 1)
            cases.append({"start": 1, "index": 0, "match": 0, "seq": seq, "pos": [[0]]})
            out["dyn_so"].append({"name": name, "tokens": toks, "max_predicts": mp, "alpha": alpha, "cases": cases})
        for npred in (4, 12, 40):
            sam = S_DynSAM(n_predicts=npred)
            sam.add_tokens(toks)
            cases = []
            for _ in range(6):
                t = toks[int(rng.integers(0, len(toks)))]
                idx, ln = sam.lookup(t)
                cases.append({"start": t, "index": idx, "match": ln, "anc": sam.to_anc(idx),
                              "seq": sam.gen_draft(idx, t)})
            cases.append({"start": 1, "index": 0, "match": 0, "anc": sam.to_anc(0), "seq": sam.gen_draft(0, 1)})
            out["dyn_s"].append({"name": name, "tokens": toks, "n_predicts": npred, "cases": cases})
    for name, (docs, eos) in corpora.items():
        flat = [t for d in docs for t in d]
        for npred in (4, 12):
            s = S_StaticSAM.build(docs, eos, False)
            s.n_predicts = npred
            cases = []
            for _ in range(8):
                p = int(rng.integers(0, len(flat) - 3))
                s.reset(); s.transfer_tokens(flat[p:p + 2])
                idx, ln = s.lookup(flat[p + 2])
                cases.append({"prefix": flat[p:p + 2], "start": flat[p + 2], "index": idx, "match": ln,
                              "seq": s.gen_draft(idx, flat[p + 2])})
            out["static_s"].append({"name": name, "n_predicts": npred, "cases": cases})
        so = quiet(SO_StaticSAM.build, docs, eos, False)
        so.device = "cpu"
        for (mp, alpha, K) in [(60, 4.0, 8), (16, 4.0, 8), (60, 4.0, 3), (30, 2.5, 1), (60, 4.0, 12)]:
            so.max_predicts, so.alpha, so.K = mp, alpha, K
            cases = []
            for _ in range(10):
                p = int(rng.integers(0, len(flat) - 4))
                plen = int(rng.integers(1, 4))
                so.reset(); so.transfer_tokens(flat[p:p + plen])
                t = flat[p + plen]
                idx, ln = so.lookup(t)
                if idx == 0:
                    continue
                for bias in (0, 1):
                    if ln - bias < 1:
                        continue
                    tree, buf = so.gen_draft(idx, ln - bias, t)
                    # recover anc_tree from the mask/pos (parent = deepest strict ancestor)
                    mask = buf["tree_attn_mask"][0, 0].int()
                    pos = buf["tree_position_ids"][0].tolist()
                    anc = []
                    for i in range(len(tree)):
                        cand = [j for j in range(len(tree)) if j != i and mask[i, j] and pos[j] == pos[i] - 1]
                        anc.append(cand[0] if cand else -1)
                    cases.append({"prefix": flat[p:p + plen], "start": t, "index": idx, "match": ln - bias,
                                  "tree": tree, "anc": anc})
            out["tree"].append({"name": name, "max_predicts": mp, "alpha": alpha, "K": K, "cases": cases})
    dump("drafts.json.gz", out)


# --------------------------------------------------------------------------- 6: buffers
def random_parent_array(rng, n, shape):
    anc = [-1]
    for i in range(1, n):
        if shape == "chain":
            anc.append(i - 1)
        elif shape == "star":
            anc.append(0)
        elif shape == "bushy":
            anc.append(int(rng.integers(max(0, i - 4), i)))
        else:
            anc.append(int(rng.integers(0, i)))
    return anc


def gen_buffers_fixture(rng):
    out = {"so": [], "token_recycle": []}
    sam = SO_StaticSAM(device="cpu")
    for n in (1, 2, 3, 7, 16, 33, 60, 64):
        for shape in ("chain", "star", "bushy", "random"):
            anc = random_parent_array(rng, n, shape)
            b = sam.gen_buffers(anc)
            out["so"].append({"anc": anc, "pos": b["tree_position_ids"].tolist(),
                              "mask": b["tree_attn_mask"][0, 0].int().tolist(),
                              "retrieve": b["tree_retrieve_indices"].tolist()})
    with open(f"{R}/samd/config/token_recycle.json") as f:
        adj = json.load(f)["tree_adj"]
    tree = [adj[str(i)] for i in range(len(adj))]
    trees = {"token_recycle.json": tree, "tiny": [[1, 2], [3], [], []], "single": [[]]}
    for name, t in trees.items():
        b = tr_gen_buffers(t, "cpu")
        out["token_recycle"].append({"name": name, "tree": t, "pos": b["tree_position_ids"].tolist(),
                                     "mask": (b["tree_attn_mask"][0, 0] != 0).int().tolist(),
                                     "mask_dtype": str(b["tree_attn_mask"].dtype),
                                     "retrieve": b["tree_retrieve_indices"].tolist()})
    dump("buffers.json.gz", out)
    return tree


# --------------------------------------------------------------------------- 7: DraftModel decisions
def gen_draft_model(rng, corpora, tr_tree):
    out = {"so": [], "s": []}
    docs, eos = corpora["vocab_docs"]
    flat = [t for d in docs for t in d]
    for (mp, alpha, K, bias) in [(60, 4.0, 8, 0), (60, 4.0, 8, 5), (16, 4.0, 8, 1), (24, 2.0, 4, 2)]:
        cfg = SO_SamdConfig(max_predicts=mp, alpha=alpha, K=K, len_bias=bias)
        so = quiet(SO_StaticSAM.build, docs, eos, False)
        d = SO_DraftModel(cfg, sam_static=so, device="cpu")
        d.reset()
        p = int(rng.integers(0, len(flat) - 80))
        prompt = flat[p:p + 30] + rng.integers(3, 40, 6).tolist() + flat[p + 10:p + 25]
        d.update(torch.tensor(prompt))
        steps = []
        for s in range(14):
            t = flat[p + 25 + s] if s % 4 != 3 else int(rng.integers(3, 40))
            ty, toks, buf = d.lookup(t)
            rec = {"start": t, "type": ty.value, "tokens": toks}
            if ty.value == "tree":
                rec["pos"] = buf["tree_position_ids"].tolist()
                rec["retrieve"] = buf["tree_retrieve_indices"].tolist()
                rec["mask"] = buf["tree_attn_mask"][0, 0].int().tolist()
            a = int(rng.integers(1, min(4, len(toks)) + 1))
            acc = toks[:a]
            d.update(torch.tensor(acc))
            rec["accepted"] = acc
            rec["cursors"] = [d.sam_dyn.cur_index, d.sam_dyn.cur_length, d.sam_static.cur_index, d.sam_static.cur_length]
            steps.append(rec)
        out["so"].append({"docs": docs, "eos": eos, "max_predicts": mp, "alpha": alpha, "K": K, "len_bias": bias,
                          "prompt": prompt, "steps": steps})
    for (npred, thr, bias, use_static) in [(12, 3, 1, True), (8, 2, 0, True), (12, 3, 1, False), (40, 5, 5, True)]:
        cfg = S_SamdConfig(n_predicts=npred, len_threshold=thr, len_bias=bias, tree_method="token_recycle", tree=tr_tree)
        st = S_StaticSAM.build(docs, eos, False) if use_static else None
        d = S_DraftModel(cfg, sam_static=st, lm=None, device="cpu")
        d.reset()
        p = int(rng.integers(0, len(flat) - 80))
        prompt = flat[p:p + 30] + rng.integers(3, 40, 6).tolist() + flat[p + 10:p + 25]
        V = 48
        plog = torch.from_numpy(perm_logits(rng, len(prompt), V))
        d.update(tokens=torch.tensor(prompt), tree_tokens=torch.tensor(prompt), tree_logits=plog)
        steps = []
        for s in range(14):
            t = flat[p + 25 + s] if s % 4 != 3 else int(rng.integers(3, 40))
            ty, toks, buf = d.lookup(t)
            logits = torch.from_numpy(perm_logits(rng, len(toks), V))
            a = int(rng.integers(1, 4))
            acc = toks[:a]
            d.update(tokens=torch.tensor(acc), tree_tokens=torch.tensor(toks), tree_logits=logits)
            steps.append({"start": t, "type": ty.value, "tokens": toks, "logits": logits.int().tolist(), "accepted": acc})
        out["s"].append({"docs": docs, "eos": eos, "n_predicts": npred, "len_threshold": thr, "len_bias": bias,
                         "use_static": use_static, "prompt": prompt, "prompt_logits": plog.int().tolist(), "vocab": V,
                         "steps": steps})
    dump("draft_model.json.gz", out)


# --------------------------------------------------------------------------- 8: posterior
def gen_posterior(rng):
    out = []
    sam = SO_StaticSAM(device="cpu")
    gcfg = SamdGenerationConfig()
    V = 24
    for case in range(40):
        n = int(rng.integers(1, 20))
        shape = ["chain", "bushy", "random", "star"][case % 4]
        anc = random_parent_array(rng, n, shape)
        tokens = rng.integers(0 if case % 5 == 0 else 1, V, n).tolist()
        logits = rng.standard_normal((n, V)).astype(np.float32)
        # steer: make many nodes' arg-max equal to one of their children's tokens
        for i in range(n):
            kids = [j for j in range(n) if anc[j] == i]
            r = rng.random()
            if kids and r < 0.75:
                logits[i, tokens[kids[int(rng.integers(0, len(kids)))]]] = 9.0
            elif r < 0.9:
                logits[i, 0] = 9.0           # arg-max = pad token 0 -> the padding quirk
            if case % 7 == 0:               # exact ties: first max wins
                logits[i, :] = np.round(logits[i, :])
        is_seq = (shape == "chain" and case % 8 == 0)
        lt = torch.from_numpy(logits)
        if is_seq:
            cand = torch.tensor([tokens])
            best, acc, sp = eval_posterior(lt[None], cand, gcfg)
            rec = {"type": "sequence", "retrieve": None}
        else:
            retrieve = sam.gen_buffers(anc)["tree_retrieve_indices"]
            tokens_ext = torch.tensor(tokens + [0])
            cand = tokens_ext[retrieve]
            best, acc, sp = eval_posterior(lt[retrieve], cand, gcfg)
            rec = {"type": "tree", "retrieve": retrieve.tolist()}
        rec.update({"anc": anc, "tokens": tokens, "logits": logits.tolist(), "candidates": cand.tolist(),
                    "best": int(best), "accept": int(acc), "next_argmax": int(sp.argmax()),
                    "accepted_tokens": cand[int(best)][:int(acc)].tolist()})
        if not is_seq:
            rec["accepted_indices"] = retrieve[int(best)][:int(acc)].tolist()
        out.append(rec)
    dump("posterior.json.gz", out)


# --------------------------------------------------------------------------- 9: token recycle
def gen_token_recycle(rng, tr_tree):
    out = []
    V = 64
    for tree in (tr_tree, [[1, 2], [3], [], []]):
        cfg = S_SamdConfig(tree_method="token_recycle", tree=tree)
        tr = TokenRecycle(cfg, None, torch.float32, "cpu")
        rounds = []
        for r in range(5):
            n = len(tree)
            toks = rng.integers(0, V, n).tolist()
            logits = perm_logits(rng, n, V)
            tr.update(tree_tokens=torch.tensor(toks), tree_logits=torch.from_numpy(logits))
            starts = [toks[0], toks[-1], int(rng.integers(0, V))]
            rounds.append({"tree_tokens": toks, "logits": logits.astype(int).tolist(),
                           "drafts": [[s, tr.gen_draft(s)[0]] for s in starts]})
        out.append({"tree": tree, "vocab": V, "rounds": rounds})
    dump("token_recycle.json.gz", out)


# --------------------------------------------------------------------------- 10: whole-loop traces
class ScriptedLM(_NpScriptedLM):
    def logits(self, committed, tokens, anc):
        return torch.from_numpy(super().logits(committed, tokens, anc))


def anc_from_buffers(tokens, buf):
    if "tree_attn_mask" not in buf:
        return [i - 1 for i in range(len(tokens))]
    mask = (buf["tree_attn_mask"][0, 0] != 0).int()
    pos = buf["tree_position_ids"][0].tolist()
    anc = []
    for i in range(len(tokens)):
        c = [j for j in range(len(tokens)) if j != i and mask[i, j] and pos[j] == pos[i] - 1]
        anc.append(c[0] if c else -1)
    return anc


def run_loop_so(draft, lm, cfg, prompt, max_new_tokens, max_cache_len, eos, device="cpu"):
    """Restates SamdModel.generate/prefill/decode/update_state (samd_sam_only/samd_model.py:96-237)
    around the IMPORTED DraftModel / gen_candidates / eval_posterior, with `lm` in place of the
    HF forward and a list in place of the KV cache (cache rows == committed token positions)."""
    gcfg = SamdGenerationConfig(max_new_tokens=max_new_tokens, max_cache_len=max_cache_len)
    draft.reset()
    ids = list(prompt)
    kv_rows = list(range(len(prompt)))          # which (step-local) row each cache slot came from: identity at prefill
    draft.update(tokens=torch.tensor(prompt))
    sample_p = lm.logits([], prompt, [i - 1 for i in range(len(prompt))])[-1:].clone()
    input_length, decode_tokens, decode_steps, acc_list, trace = len(prompt), 0, 0, [], []
    for _ in range(max_new_tokens):
        if input_length + decode_tokens + cfg.max_predicts >= max_cache_len:
            break
        cands = gen_candidates(sample_p, None, draft, cfg, gcfg, device)
        tokens = cands.tokens[0].tolist()
        anc = anc_from_buffers(tokens, cands.buffers_kwargs)
        logits = lm.logits(ids, tokens, anc)
        if cands.type.value == "sequence":
            cl, retrieve = logits[None], None
        else:
            retrieve = cands.buffers_kwargs["tree_retrieve_indices"]
            cl = logits[retrieve]
        best, acc, sample_p = eval_posterior(cl, cands.candidate_tokens, gcfg)
        new = cands.candidate_tokens[best][:acc]
        idx = None if retrieve is None else retrieve[best][:acc].tolist()
        draft.update(tokens=new)
        new_ids = new.tolist()
        full_new = list(new_ids)
        eos_index = None
        if eos in new_ids:
            eos_index = new_ids.index(eos); new_ids = new_ids[:eos_index + 1]
        ids.extend(new_ids)
        decode_steps += 1; decode_tokens += len(new_ids); acc_list.append(len(new_ids))
        trace.append({"type": cands.type.value, "tokens": tokens, "anc": anc, "best": int(best), "accept": int(acc),
                      "accepted": full_new, "kv_indices": idx, "node_argmax": logits.argmax(-1).tolist()})
        if eos_index is not None or decode_tokens >= max_new_tokens:
            break
    return {"output_ids": ids[:input_length + max_new_tokens], "decode_tokens": decode_tokens,
            "decode_steps": decode_steps, "accept_lengths": acc_list, "trace": trace}


def gen_loop_so(rng):
    out = []
    V = 96
    docs = [stream_markov(rng, 80, vocab=V, succ=2, noise=0.02) for _ in range(16)] + [[i] for i in range(V)]
    flat = [t for d in docs[:16] for t in d]
    for case, (mp, alpha, K, bias, max_new, cache_len, eos_at) in enumerate(
            [(16, 4.0, 8, 0, 64, 512, None), (60, 4.0, 8, 0, 96, 512, None), (16, 4.0, 8, 2, 48, 512, 30),
             (8, 2.0, 3, 0, 40, 100, None), (60, 4.0, 8, 5, 64, 2048, None)]):
        cfg = SO_SamdConfig(max_predicts=mp, alpha=alpha, K=K, len_bias=bias)
        so = quiet(SO_StaticSAM.build, docs, 2, False)
        d = SO_DraftModel(cfg, sam_static=so, device="cpu")
        p = int(rng.integers(0, len(flat) - 400))
        prompt = flat[p:p + 24] + rng.integers(3, V, 4).tolist() + flat[p + 8:p + 20]
        # the "true" continuation mixes corpus spans, prompt repeats and noise
        cont = []
        while len(cont) < max_new + 8:
            r = rng.random()
            if r < 0.5:
                q = int(rng.integers(0, len(flat) - 30)); cont += flat[q:q + int(rng.integers(4, 24))]
            elif r < 0.75:
                q = int(rng.integers(0, len(prompt) - 6)); cont += prompt[q:q + int(rng.integers(3, 10))]
            else:
                cont += rng.integers(3, V, int(rng.integers(1, 4))).tolist()
        cont = [t if t != 2 else 3 for t in cont]
        if eos_at is not None:
            cont[eos_at] = 2
        lm = ScriptedLM(prompt + cont, V)
        res = run_loop_so(d, lm, cfg, prompt, max_new, cache_len, 2)
        out.append({"docs": docs, "eos": 2, "vocab": V, "max_predicts": mp, "alpha": alpha, "K": K, "len_bias": bias,
                    "max_new_tokens": max_new, "max_cache_len": cache_len, "prompt": prompt,
                    "target": prompt + cont, **res})
        print(f"  loop_so[{case}]: steps={res['decode_steps']} tokens={res['decode_tokens']} "
              f"types={[t['type'][0] for t in res['trace']]}")
    dump("loop_so.json.gz", out)


def run_loop_s(draft, lm, cfg, prompt, max_new_tokens, max_cache_len, eos, base_buffers, vocab):
    """Restates samd/samd_model.py prefill (:101-128), decode (:131-182), update_state (:185-211) and
    generate (:230-274) for tree_method=token_recycle around the IMPORTED samd DraftModel."""
    gcfg = SamdGenerationConfig(max_new_tokens=max_new_tokens, max_cache_len=max_cache_len)
    draft.reset()
    ids = list(prompt)
    plog = lm.logits([], prompt, [i - 1 for i in range(len(prompt))])
    draft.update(tokens=torch.tensor(prompt), last_hidden_states=None,
                 tree_tokens=torch.tensor(prompt), tree_logits=plog)
    sample_p = plog[-1:].clone()
    base_anc = anc_from_buffers(list(range(base_buffers["tree_position_ids"].shape[1])), base_buffers)
    base_retrieve = base_buffers["tree_retrieve_indices"]
    input_length, decode_tokens, decode_steps, acc_list, trace = len(prompt), 0, 0, [], []
    for _ in range(max_new_tokens):
        if input_length + decode_tokens + cfg.max_predicts >= max_cache_len:
            break
        cands = gen_candidates(sample_p, base_retrieve, draft, cfg, gcfg, "cpu")
        tokens = cands.tokens[0].tolist()
        if cands.type.value == "sequence":
            anc = [i - 1 for i in range(len(tokens))]
            logits = lm.logits(ids, tokens, anc)
            cl, retrieve = logits[None], None
        else:
            anc = base_anc
            logits = lm.logits(ids, tokens, anc)
            retrieve = base_retrieve
            cl = logits[retrieve]
        best, acc, sample_p = eval_posterior(cl, cands.candidate_tokens, gcfg)
        new = cands.candidate_tokens[best][:acc]
        idx = None if retrieve is None else retrieve[best][:acc].tolist()
        draft.update(tokens=new, last_hidden_states=None, tree_tokens=cands.tokens[0], tree_logits=logits)
        new_ids = new.tolist()
        full_new = list(new_ids)
        eos_index = None
        if eos in new_ids:
            eos_index = new_ids.index(eos); new_ids = new_ids[:eos_index + 1]
        ids.extend(new_ids)
        decode_steps += 1; decode_tokens += len(new_ids); acc_list.append(len(new_ids))
        trace.append({"type": cands.type.value, "tokens": tokens, "best": int(best), "accept": int(acc),
                      "accepted": full_new, "kv_indices": idx, "node_argmax": logits.argmax(-1).tolist()})
        if eos_index is not None or decode_tokens >= max_new_tokens:
            break
    return {"output_ids": ids[:input_length + max_new_tokens], "decode_tokens": decode_tokens,
            "decode_steps": decode_steps, "accept_lengths": acc_list, "trace": trace}


def gen_loop_s(rng, tr_tree):
    out = []
    V = 96
    docs = [stream_markov(rng, 80, vocab=V, succ=2, noise=0.02) for _ in range(16)] + [[i] for i in range(V)]
    flat = [t for d in docs[:16] for t in d]
    for case, (npred, thr, bias, use_static, max_new) in enumerate(
            [(12, 3, 1, True, 64), (8, 2, 0, False, 48), (40, 5, 5, True, 64)]):
        cfg = S_SamdConfig(n_predicts=npred, len_threshold=thr, len_bias=bias, tree_method="token_recycle", tree=tr_tree)
        st = S_StaticSAM.build(docs, 2, False) if use_static else None
        d = S_DraftModel(cfg, sam_static=st, lm=None, device="cpu")
        base = d.tree_model.gen_buffers()
        p = int(rng.integers(0, len(flat) - 400))
        prompt = flat[p:p + 24] + rng.integers(3, V, 4).tolist() + flat[p + 8:p + 20]
        cont = []
        while len(cont) < max_new + 8:
            r = rng.random()
            if r < 0.5:
                q = int(rng.integers(0, len(flat) - 30)); cont += flat[q:q + int(rng.integers(4, 24))]
            elif r < 0.75:
                q = int(rng.integers(0, len(prompt) - 6)); cont += prompt[q:q + int(rng.integers(3, 10))]
            else:
                cont += rng.integers(3, V, int(rng.integers(1, 4))).tolist()
        cont = [t if t != 2 else 3 for t in cont]
        lm = ScriptedLM(prompt + cont, V)
        res = run_loop_s(d, lm, cfg, prompt, max_new, 512, 2, base, V)
        out.append({"docs": docs, "eos": 2, "vocab": V, "n_predicts": npred, "len_threshold": thr, "len_bias": bias,
                    "max_predicts": cfg.max_predicts, "use_static": use_static, "tree": tr_tree,
                    "max_new_tokens": max_new, "max_cache_len": 512,
                    "prompt": prompt, "target": prompt + cont, **res})
        print(f"  loop_s[{case}]: steps={res['decode_steps']} tokens={res['decode_tokens']} "
              f"types={[t['type'][0] for t in res['trace']]}")
    dump("loop_s.json.gz", out)


def main():
    rng = np.random.default_rng(20250103)
    corpora = gen_sam_traces(rng)
    gen_drafts(rng, corpora)
    tr_tree = gen_buffers_fixture(rng)
    gen_draft_model(rng, corpora, tr_tree)
    gen_posterior(rng)
    gen_token_recycle(rng, tr_tree)
    gen_loop_so(rng)
    gen_loop_s(rng, tr_tree)


if __name__ == "__main__":
    main()
