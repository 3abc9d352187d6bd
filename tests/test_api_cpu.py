"""CPU checks of the drop-in surface: the packages export the reference's names with the reference's signatures and
defaults (samd_sam_only/__init__.py:1-5, samd/__init__.py:1-5), host-side logic (configs, generation config, tree
tables, static-automaton files) works without a GPU, and anything that needs the device fails loudly."""
import inspect

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import samd_hip


def test_so_surface_and_defaults():
    import samd_sam_only as SO
    for name in ("SamdConfig", "SamdModel", "SamdGenerationConfig", "DraftModel", "build_sam", "load_sam", "dump_sam"):
        assert hasattr(SO, name)
    c = SO.SamdConfig()
    assert (c.max_predicts, c.alpha, c.K, c.len_bias, c.cache_type) == (60, 4.0, 8, 5, "static")
    g = SO.SamdGenerationConfig()
    assert (g.max_steps, g.max_new_tokens, g.max_cache_len, g.greedy, g.temperature, g.top_p, g.top_k) == (512, 512, 2048, True, 0.0, 0.0, 0)
    assert list(inspect.signature(SO.SamdModel.__init__).parameters)[1:] == ["samd_config", "lm", "draft", "eos_token_id", "dtype", "device", "stop_token_id"]
    assert list(inspect.signature(SO.DraftModel.__init__).parameters)[1:7] == ["config", "sam_dyn", "sam_static", "lm", "dtype", "device"]
    assert list(inspect.signature(SO.SamdModel.generate).parameters)[1:] == ["input_ids", "attention_mask", "generation_config"]
    from samd_sam_only.sam import DynSAM, StaticSAM
    assert list(inspect.signature(DynSAM.__init__).parameters)[1:4] == ["max_predicts", "alpha", "device"]
    assert list(inspect.signature(StaticSAM.__init__).parameters)[1:] == ["max_predicts", "alpha", "K", "device"]
    from samd_sam_only.samd_config import ForwardType
    assert [e.value for e in ForwardType] == ["prefill", "seq_decode", "tree_decode"]
    with pytest.warns(RuntimeWarning, match="capped at 128"):
        SO.SamdConfig(max_predicts=129)                    # one wavefront builds one draft, two nodes per lane: capped, not refused


def test_draft_size_limit_is_a_warning_not_an_error():
    """max_predicts / n_predicts up to 128 are served as the reference serves them (round 5; 64 before); above that -- the reference accepts
    any value -- this implementation caps drafts at 128 nodes (the value itself is kept: the cache guard of generate() uses it as the
    reference does) and says so once per config"""
    import warnings
    import samd_sam_only as SO
    import samd as S
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert SO.SamdConfig(max_predicts=200).max_predicts == 200
        assert S.SamdConfig(n_predicts=130).n_predicts == 130
        assert SO.SamdConfig(max_predicts=128).max_predicts == 128 and S.SamdConfig(n_predicts=100).n_predicts == 100      # no warning: served exactly
    assert len(w) == 2 and all("capped at 128" in str(x.message) for x in w)
    with pytest.raises(ValueError):
        SO.SamdConfig(max_predicts=0)


def test_draft_size_follows_what_the_verifier_can_run():
    """a runner without row-major matrices (or in an attention mode without the two-tile form) verifies at most 64 rows; the
    session's parameters are clamped to that when the engine is made -- with a warning -- instead of failing inside forward_rows on the
    first wide draft.  Host logic only: no GPU needed."""
    import warnings
    from samd_sam_only.sam._common import clamp_to_verifier, s_params, so_params

    class Verifier:
        def __init__(self, cap):
            self.cap = cap

        def max_draft_rows(self):
            return self.cap

    class Draft:
        pass
    assert so_params(128).max_predicts == 128 and so_params(128, cap=64).max_predicts == 64 and so_params(40, cap=64).max_predicts == 40
    assert s_params(128).n_predicts == 128 and s_params(100, cap=64).n_predicts == 64 and so_params(500, cap=None).max_predicts == samd_hip.MAX_DRAFT
    d = Draft()
    with pytest.warns(RuntimeWarning, match="capped at 64"):
        assert clamp_to_verifier(d, Verifier(64), 100, "max_predicts") == 64
    assert d.draft_cap == 64
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert clamp_to_verifier(d, Verifier(64), 100, "max_predicts") == 64      # said once per model and cap
        assert clamp_to_verifier(d, Verifier(128), 100, "max_predicts") == 128    # nothing to say: the verifier can run it
        assert clamp_to_verifier(Draft(), object(), 100, "max_predicts") == samd_hip.MAX_DRAFT   # a verifier that says nothing: the library's limit
    from samd_hip.llama import LlamaRunner
    r = LlamaRunner.__new__(LlamaRunner)
    for released, mode, head, want in ((False, "split", False, 128), (True, "split", False, 64), (False, "block", False, 64), (False, "split2", False, 64),
                                       (False, "split3", False, 128), (False, "split", True, 64)):
        r.row_major_released, r.attention, r.draft_head = released, mode, head
        assert r.max_draft_rows() == want, (released, mode, head)


def test_s_surface_and_defaults():
    import samd as S
    c = S.SamdConfig()
    assert (c.n_predicts, c.max_predicts, c.len_threshold, c.len_bias, c.tree_method) == (40, 70, 5, 5, "token_recycle")
    assert len(c.tree) == 61 and [len(x) for x in c.tree[:4]] == [7, 6, 5, 3]
    levels = {0: 0}
    for node, childs in enumerate(c.tree):
        for ch in childs:
            levels[ch] = levels[node] + 1
    assert [list(levels.values()).count(d) for d in range(6)] == [1, 7, 20, 21, 8, 4]
    from samd.sam import DynSAM, NullStaticSAM, StaticSAM          # noqa: F401
    from samd.tree_model import TokenRecycle, TreeModel, tree_model_cls
    assert tree_model_cls["token_recycle"] is TokenRecycle and issubclass(TokenRecycle, TreeModel)
    assert list(inspect.signature(S.DraftModel.__init__).parameters)[1:8] == ["config", "sam_dyn", "sam_static", "tree_model", "lm", "dtype", "device"]


def test_sampling_config_builds_processors():
    import samd_sam_only as SO
    g = SO.SamdGenerationConfig(greedy=False, temperature=0.7, top_p=0.9, top_k=20)
    assert len(g.logits_processor) == 3
    with pytest.raises(AssertionError):
        SO.SamdGenerationConfig(greedy=False, temperature=0.0)


def test_static_sam_files_and_views(tmp_path):
    """build_sam / dump_sam / load_sam and the host-side views are CPU work (the builder is native host code)."""
    import samd_sam_only as SO
    from oracle import sam_oracle as O
    rng = np.random.default_rng(1)
    docs = [rng.integers(3, 30, 40).tolist() for _ in range(5)] + [[i] for i in range(30)]
    sam = SO.build_sam(docs, 2)
    ora = O.StaticSAM.build(docs, 2)
    e = ora.export()
    st = sam.states
    assert [s.link for s in st] == e["link"].tolist()
    assert [s.cnt_endpos for s in st] == e["aux"].tolist()
    tok, dst, _ = ora.export_topk()
    topk = sam.states_topk_next
    for i in range(len(st)):
        k = len(topk[i])
        assert topk[i] == list(zip(tok[i][:k].tolist(), dst[i][:k].tolist()))
    p = str(tmp_path / "a.sam")
    SO.dump_sam(p, sam)
    again = SO.load_sam(p)
    assert [s.next for s in again.states] == [s.next for s in st]


def test_reference_style_pickle_loads(tmp_path):
    """a pickle shaped like the reference's dump_sam output (object with `states` = list of SAMState) converts on load."""
    import pickle
    import samd_sam_only as SO
    from samd_sam_only.sam.static_sam import StaticSAM
    docs = [[5, 6, 7, 5, 6, 8], [9, 5, 6, 7]] + [[i] for i in range(10)]
    sam = SO.build_sam(docs, 2)
    states = sam.states
    # what pickle stores for a plain reference object: the class reference + __dict__
    blob = pickle.dumps({"states": states, "max_predicts": 40, "alpha": 4.0, "K": 8, "device": "cuda", "cur_index": 3})
    fake = StaticSAM.__new__(StaticSAM)
    fake.__setstate__(pickle.loads(blob))
    assert [s.next for s in fake.states] == [s.next for s in states]
    assert fake.states_topk_next == sam.states_topk_next


def test_device_ops_fail_loudly_without_gpu():
    if samd_hip.lib().samd_device_count() > 0:
        pytest.skip("a GPU is visible")
    import samd_sam_only as SO
    sam = SO.build_sam([[1, 2, 3]], 2)
    with pytest.raises(samd_hip.SamdError):
        sam.lookup(1)
    with pytest.raises(samd_hip.SamdError):
        SO.DraftModel(SO.SamdConfig(), sam_static=sam).reset()
    with pytest.raises(samd_hip.SamdError):
        from samd_hip.llama import LlamaRunner
        LlamaRunner.random_init(dict(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, vocab_size=64), 64)
