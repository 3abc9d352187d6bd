"""Pins the CPU oracle (oracle/sam_oracle.c) against fixtures produced by the imported Python
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import sam_oracle as O
from scripted_lm import ScriptedLM


def table_of(sam):
    e = sam.export()
    edges, k = [], 0
    for d in e["deg"]:
        edges.append([[int(t), int(x)] for t, x in zip(e["edge_tok"][k:k + d], e["edge_dst"][k:k + d])])
        k += d
    return {"link": e["link"].tolist(), "length": e["length"].tolist(), "aux": e["aux"].tolist(), "edges": edges}, e


def test_dyn_traces(golden):
    for case in golden("sam_traces.json.gz")["dyn"]:
        sam = O.DynSAM()
        toks, cuts = case["tokens"], case["cuts"]
        for (a, b), cur, probes in zip(zip(cuts[:-1], cuts[1:]), case["cursors"], case["probes"]):
            sam.add_tokens(toks[a:b])
            assert list(sam.cursor()) == cur, case["name"]
            for t, pi, pl in probes:
                assert sam.lookup(t) == (pi, pl)
        tab, e = table_of(sam)
        assert tab == case["table"], case["name"]      # incl. edges in dict (insertion) order
        assert e["text"].tolist() == case["input_ids"]
        assert (sam.last, sam.max_length) == (case["last"], case["max_length"])


def test_dyn_reset(golden):
    case = golden("sam_traces.json.gz")["dyn"][1]
    sam = O.DynSAM()
    sam.add_tokens(case["tokens"][:50])
    sam.reset()
    assert sam.num_states == 1 and sam.cursor() == (0, 0)
    sam.add_tokens(case["tokens"])
    assert table_of(sam)[0] == case["table"]


def test_static_so_traces(golden):
    for case in golden("sam_traces.json.gz")["static_so"]:
        sam = O.StaticSAM.build(case["docs"], case["eos"])
        tab, _ = table_of(sam)
        assert tab == case["table"], case["name"]
        tok, dst, n = sam.export_topk()
        got = [[[int(t), int(d)] for t, d in zip(tok[i, :n[i]], dst[i, :n[i]])] for i in range(len(n))]
        assert got == case["topk"], case["name"]
        sam.reset()
        for t, w in zip(case["query"], case["walk"]):
            assert list(sam.lookup(t)) == w
            sam.transfer_tokens([t])
            assert list(sam.cursor()) == w
        assert (sam.last, sam.max_length) == (case["last"], case["max_length"])


def test_static_s_traces(golden):
    for case in golden("sam_traces.json.gz")["static_s"]:
        sam = O.StaticSAMFull.build(case["docs"], case["eos"])
        tab, e = table_of(sam)
        assert tab == case["table"], case["name"]
        assert e["text"].tolist() == case["input_ids"]
        sam.reset()
        for t, w in zip(case["query"], case["walk"]):
            sam.transfer_tokens([t])
            assert list(sam.cursor()) == w


def test_dyn_drafts(golden):
    g = golden("drafts.json.gz")
    for case in g["dyn_so"]:
        sam = O.DynSAM(case["max_predicts"], case["alpha"])
        sam.add_tokens(case["tokens"])
        for c in case["cases"]:
            if c["index"] != 0:
                assert sam.lookup(c["start"]) == (c["index"], c["match"])
            assert sam.gen_draft(c["index"], c["match"], c["start"]) == c["seq"]
            assert c["pos"] == [list(range(len(c["seq"])))]
    for case in g["dyn_s"]:
        sam = O.DynSAM(n_predicts=case["n_predicts"])
        sam.add_tokens(case["tokens"])
        for c in case["cases"]:
            assert sam.to_anc(c["index"]) == c["anc"]
            assert sam.gen_draft_fixed(c["index"], c["start"]) == c["seq"]


def test_static_s_drafts(golden):
    g = golden("drafts.json.gz")
    corp = {c["name"]: c for c in golden("sam_traces.json.gz")["static_s"]}
    corp.update(g.get("corpora", {}))                        # (the wide fixtures carry their corpora inline)
    for case in g["static_s"]:
        sam = O.StaticSAMFull.build(corp[case["name"]]["docs"], corp[case["name"]]["eos"])
        sam.n_predicts = case["n_predicts"]
        for c in case["cases"]:
            sam.reset()
            sam.transfer_tokens(c["prefix"])
            assert sam.lookup(c["start"]) == (c["index"], c["match"])
            assert sam.gen_draft(c["index"], c["start"]) == c["seq"]


def test_tree_drafts(golden):
    g = golden("drafts.json.gz")
    corp = {c["name"]: c for c in golden("sam_traces.json.gz")["static_so"]}
    corp.update(g.get("corpora", {}))
    ncases = 0
    for case in g["tree"]:
        sam = O.StaticSAM.build(corp[case["name"]]["docs"], corp[case["name"]]["eos"])
        sam.max_predicts, sam.alpha, sam.K = case["max_predicts"], case["alpha"], case["K"]
        for c in case["cases"]:
            tree, anc = sam.gen_draft_tree(c["index"], c["match"], c["start"])
            assert tree == c["tree"] and anc == c["anc"], (case["name"], c)
            ncases += 1
    assert ncases > 50


def test_gen_buffers(golden):
    g = golden("buffers.json.gz")
    for c in g["so"]:
        b = O.gen_buffers(c["anc"])
        assert b["tree_position_ids"].tolist() == c["pos"]
        assert b["tree_attn_mask"][0, 0].astype(int).tolist() == c["mask"]
        assert b["tree_retrieve_indices"].tolist() == c["retrieve"]
    for c in g["token_recycle"]:
        b = O.tr_gen_buffers(c["tree"])
        assert b["tree_position_ids"].tolist() == c["pos"]
        assert (b["tree_attn_mask"][0, 0] != 0).astype(int).tolist() == c["mask"]
        assert b["tree_retrieve_indices"].tolist() == c["retrieve"]
        assert c["mask_dtype"] == "torch.float32"


def test_draft_model_so(golden):
    for case in golden("draft_model.json.gz")["so"]:
        st = O.StaticSAM.build(case["docs"], case["eos"])
        d = O.DraftModel(case["max_predicts"], case["alpha"], case["K"], case["len_bias"], sam_static=st)
        d.reset()
        d.update(case["prompt"])
        for s in case["steps"]:
            ty, toks, buf = d.lookup(s["start"])
            assert ty == s["type"] and toks == s["tokens"]
            if ty == "tree":
                assert buf["tree_position_ids"].tolist() == s["pos"]
                assert buf["tree_retrieve_indices"].tolist() == s["retrieve"]
                assert buf["tree_attn_mask"][0, 0].astype(int).tolist() == s["mask"]
            d.update(s["accepted"])
            assert [*d.sam_dyn.cursor(), *d.sam_static.cursor()] == s["cursors"]


def s_lookup(dyn, st, tr, start, n_predicts, len_threshold, len_bias):
    """samd/draft.py:52-63 on the oracle pieces."""
    idd, md = dyn.lookup(start)
    if st is not None:
        ist, ms = st.lookup(start)
    else:
        ist, ms = 0, 0          # NullStaticSAM never leaves the root (S/sam/static_sam.py:128-137)
    ms -= len_bias
    if max(md, ms) >= len_threshold:
        if md >= ms:
            return "sequence", dyn.gen_draft_fixed(idd, start)
        return "sequence", st.gen_draft(ist, start)
    return "tree", tr.gen_draft(start)


def test_draft_model_s(golden):
    tr_tree = golden("buffers.json.gz")["token_recycle"][0]["tree"]
    for case in golden("draft_model.json.gz")["s"]:
        dyn = O.DynSAM(n_predicts=case["n_predicts"])
        st = O.StaticSAMFull.build(case["docs"], case["eos"]) if case["use_static"] else None
        if st is not None:
            st.n_predicts = case["n_predicts"]
        tr = O.TokenRecycle(tr_tree, case["vocab"])
        dyn.add_tokens(case["prompt"])
        if st is not None:
            st.transfer_tokens(case["prompt"])
        tr.update(case["prompt"], O.topk8_rows(np.asarray(case["prompt_logits"], np.float32)))
        for s in case["steps"]:
            ty, toks = s_lookup(dyn, st, tr, s["start"], case["n_predicts"], case["len_threshold"], case["len_bias"])
            assert ty == s["type"] and toks == s["tokens"]
            dyn.add_tokens(s["accepted"])
            if st is not None:
                st.transfer_tokens(s["accepted"])
            tr.update(toks, O.topk8_rows(np.asarray(s["logits"], np.float32)))


def test_posterior(golden):
    for c in golden("posterior.json.gz"):
        logits = np.asarray(c["logits"], np.float32)
        am = O.argmax_rows(logits)
        if c["type"] == "sequence":
            best, acc, nn = O.eval_posterior(am, c["tokens"], None)
            cand = np.asarray([c["tokens"]])
        else:
            ret = np.asarray(c["retrieve"], np.int64)
            assert O.gen_buffers(c["anc"])["tree_retrieve_indices"].tolist() == c["retrieve"]
            cand = O.candidates(c["tokens"], ret)
            assert cand.tolist() == c["candidates"]
            best, acc, nn = O.eval_posterior(am, c["tokens"], ret)
            idx = ret[best][:acc].tolist()
            assert idx == c["accepted_indices"]
        assert (best, acc) == (c["best"], c["accept"])
        assert int(am[nn]) == c["next_argmax"]
        assert cand[best][:acc].tolist() == c["accepted_tokens"]


def test_token_recycle(golden):
    for case in golden("token_recycle.json.gz"):
        tr = O.TokenRecycle(case["tree"], case["vocab"])
        for r in case["rounds"]:
            tr.update(r["tree_tokens"], O.topk8_rows(np.asarray(r["logits"], np.float32)))
            for start, draft in r["drafts"]:
                assert tr.gen_draft(start) == draft


def oracle_generate_so(case):
    """samd_sam_only/samd_model.py:193-237 (generate) with the oracle draft model and the scripted LM."""
    lm = ScriptedLM(case["target"], case["vocab"])
    st = O.StaticSAM.build(case["docs"], case["eos"])
    d = O.DraftModel(case["max_predicts"], case["alpha"], case["K"], case["len_bias"], sam_static=st)
    d.reset()
    prompt = case["prompt"]
    ids = list(prompt)
    d.update(prompt)
    start = int(O.argmax_rows(lm.logits([], prompt, [i - 1 for i in range(len(prompt))])[-1:])[0])
    dt, ds, acc_list, trace = 0, 0, [], []
    for _ in range(case["max_new_tokens"]):
        if len(prompt) + dt + case["max_predicts"] >= case["max_cache_len"]:
            break
        ty, toks, anc = d.lookup_raw(start)
        am = O.argmax_rows(lm.logits(ids, toks, anc))
        ret = None if ty == 0 else O.gen_buffers(anc)["tree_retrieve_indices"]
        best, a, nn = O.eval_posterior(am, toks, ret)
        cand = np.asarray([toks]) if ret is None else O.candidates(toks, ret)
        new = cand[best][:a].tolist()
        d.update(new)
        start = int(am[nn])
        full = list(new)
        stop = False
        if case["eos"] in new:
            new = new[:new.index(case["eos"]) + 1]
            stop = True
        ids.extend(new)
        ds += 1; dt += len(new); acc_list.append(len(new))
        trace.append({"type": "sequence" if ty == 0 else "tree", "tokens": toks, "anc": anc, "best": best, "accept": a,
                      "accepted": full, "kv_indices": None if ret is None else ret[best][:a].tolist(),
                      "node_argmax": am.tolist()})
        if stop or dt >= case["max_new_tokens"]:
            break
    return {"output_ids": ids[:len(prompt) + case["max_new_tokens"]], "decode_tokens": dt, "decode_steps": ds,
            "accept_lengths": acc_list, "trace": trace}


def test_loop_so(golden):
    for case in golden("loop_so.json.gz"):
        got = oracle_generate_so(case)
        for k in ("output_ids", "decode_tokens", "decode_steps", "accept_lengths"):
            assert got[k] == case[k], k
        for g, w in zip(got["trace"], case["trace"]):
            assert g == w
        # losslessness: the speculative output equals the scripted greedy continuation
        n = len(got["output_ids"])
        assert got["output_ids"] == case["target"][:n]


def oracle_generate_s(case):
    """samd/samd_model.py prefill (:101-128) / decode (:131-182) / update_state (:185-211) / generate (:230-274) for
    tree_method=token_recycle, on the oracle pieces and the scripted LM."""
    lm = ScriptedLM(case["target"], case["vocab"])
    tree = case["tree"]
    dyn = O.DynSAM(n_predicts=case["n_predicts"])
    st = O.StaticSAMFull.build(case["docs"], case["eos"]) if case["use_static"] else None
    if st is not None:
        st.n_predicts = case["n_predicts"]
        st.reset()
    tr = O.TokenRecycle(tree, case["vocab"])
    base = O.tr_gen_buffers(tree)
    base_anc, base_ret = base["anc_tree"].tolist(), base["tree_retrieve_indices"]
    prompt = case["prompt"]
    ids = list(prompt)
    plog = lm.logits([], prompt, [i - 1 for i in range(len(prompt))])
    dyn.add_tokens(prompt)
    if st is not None:
        st.transfer_tokens(prompt)
    tr.update(prompt, O.topk8_rows(plog))
    start = int(O.argmax_rows(plog[-1:])[0])
    dt, ds, acc_list, kinds = 0, 0, [], []
    for _ in range(case["max_new_tokens"]):
        if len(prompt) + dt + case["max_predicts"] >= case["max_cache_len"]:
            break
        ty, toks = s_lookup(dyn, st, tr, start, case["n_predicts"], case["len_threshold"], case["len_bias"])
        if ty == "sequence":
            anc, ret = [i - 1 for i in range(len(toks))], None
        else:
            anc, ret = base_anc, base_ret
        logits = lm.logits(ids, toks, anc)
        am = O.argmax_rows(logits)
        best, a, nn = O.eval_posterior(am, toks, ret)
        cand = np.asarray([toks]) if ret is None else O.candidates(toks, ret)
        new = cand[best][:a].tolist()
        dyn.add_tokens(new)
        if st is not None:
            st.transfer_tokens(new)
        tr.update(toks, O.topk8_rows(logits))
        start = int(am[nn])
        stop = False
        if case["eos"] in new:
            new = new[:new.index(case["eos"]) + 1]
            stop = True
        ids.extend(new)
        ds += 1; dt += len(new); acc_list.append(len(new)); kinds.append(ty)
        if stop or dt >= case["max_new_tokens"]:
            break
    return {"output_ids": ids[:len(prompt) + case["max_new_tokens"]], "decode_tokens": dt, "decode_steps": ds,
            "accept_lengths": acc_list, "types": kinds}


def test_loop_s(golden):
    """the oracle's full-variant loop reproduces the traces recorded from the imported reference."""
    for case in golden("loop_s.json.gz"):
        got = oracle_generate_s(case)
        for k in ("output_ids", "decode_tokens", "decode_steps", "accept_lengths"):
            assert got[k] == case[k], k
        assert got["types"] == [t["type"] for t in case["trace"]]
