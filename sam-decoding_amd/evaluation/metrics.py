"""speed / equal (reference: evaluation/speed.py:7-69, evaluation/equal.py:6-54)."""
import json
from typing import Callable, Optional, Union

import numpy as np

MT_BENCH = ["writing", "roleplay", "reasoning", "math", "coding", "extraction", "stem", "humanities"]


def _load(path: str, task: Optional[str]):
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            if not line.strip():
                continue
            obj = json.loads(line)
            if task in (None, "overall") or (task == "mt_bench" and obj["category"] in MT_BENCH) or obj["category"] == task:
                rows.append(obj)
    return rows


def speed(jsonl_file: str, jsonl_file_base: str, tokenizer: Union[str, Callable[[str], int], object], task: Optional[str] = "overall",
          report: bool = True):
    """per-question tokens/s = sum(new_tokens) / sum(wall_time) for the method; for the baseline the tokens are recounted
    by re-tokenising the answers minus BOS (speed.py:50-55); speed-up = mean / mean; MAT = mean of all accept lengths.
    `tokenizer`: a HF tokenizer name/object, or a callable text -> number of tokens incl. BOS."""
    if isinstance(tokenizer, str):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(tokenizer)
    count = tokenizer if callable(tokenizer) and not hasattr(tokenizer, "encode") else (lambda text: len(tokenizer(text).input_ids))
    speeds, accept = [], []
    for d in _load(jsonl_file, task):
        c = d["choices"][0]
        speeds.append(sum(c["new_tokens"]) / sum(c["wall_time"]))
        accept.extend(c["accept_lengths"])
    speeds0 = []
    for d in _load(jsonl_file_base, task):
        c = d["choices"][0]
        speeds0.append(sum(count(t) - 1 for t in c["turns"]) / sum(c["wall_time"]))
    tps, tps0 = float(np.mean(speeds)), float(np.mean(speeds0))
    if report:
        print("=" * 30, "Task: ", task, "=" * 30)
        print("#Mean accepted tokens: ", np.mean(accept) if accept else float("nan"))
        print("Tokens per second: ", tps)
        print("Tokens per second for the baseline: ", tps0)
        print("Speedup ratio: ", tps / tps0)
    return tps, tps0, tps / tps0, accept


def equal(jsonl_a: str, jsonl_b: str, report: bool = True) -> bool:
    """the reference's losslessness check: the answers (choices[0].turns) of two runs, compared question by question."""
    a = [str(d["choices"][0]["turns"]) for d in _load(jsonl_a, None)]
    b = [str(d["choices"][0]["turns"]) for d in _load(jsonl_b, None)]
    neq = sum(1 for x, y in zip(a, b) if x != y)
    if report:
        n = max(1, min(len(a), len(b)))
        print(f"neq: {neq}, all: {n}, ratio: {neq / n}")
        print("Result totally Equal!" if neq == 0 else "Not Equal!")
    return neq == 0
