"""The wide prefill's row-split rule (samd_hip/llama.py `_pf_split`): host logic only -- which prompt lengths a measured plan splits, and where."""
from samd_hip.llama import LlamaRunner


class Stub:
    PF_SPLIT_MIN_ROWS = LlamaRunner.PF_SPLIT_MIN_ROWS


def test_split_rule_follows_the_plan_in_64_row_steps():
    s = Stub()
    s._pf_plan = {"wgu": {1280: {64: True, 128: True, 192: False, 256: False}, 1536: {64: False, 128: False, 192: False, 256: True}}}
    f = lambda M, key="wgu": LlamaRunner._pf_split(s, key, M)
    assert [f(M) for M in (1, 512, 1024)] == [0, 0, 0]                       # short prompts: never
    assert [f(M) for M in (1025, 1280)] == [0, 0]                            # no entry for R = 1024
    assert [f(M) for M in (1281, 1344, 1345, 1408)] == [1280] * 4            # remainders of 1..128 rows
    assert [f(M) for M in (1409, 1536)] == [0, 0]                            # 129..256 rows: the plan says one call
    assert [f(M) for M in (1537, 1728, 1729, 1792)] == [0, 0, 1536, 1536]
    assert f(1300, "wqkv") == 0                                              # a projection the plan does not name


def test_an_empty_plan_means_one_call_everywhere():
    s = Stub()
    s._pf_plan = {}
    assert all(LlamaRunner._pf_split(s, k, M) == 0 for k in ("wqkv", "wo", "wgu", "wdown") for M in (1025, 1500, 2048, 4096))
