"""run_eval / get_model_answers counterpart (reference: evaluation/eval_vicuna.py:20-258, eval_llama3.py).

Per question and turn: build the prompt from the conversation template, tokenize, time `forward_func` between device
synchronisations, strip the answer at stop tokens / stop string / special tokens, and append a json line with
`turns, decoding_steps, new_tokens, wall_time, accept_lengths` -- the fields evaluation/speed.py consumes.  Three
warm-up passes over the first question precede the measurement (eval_vicuna.py:92-156).  With torch.distributed
initialised every rank evaluates the contiguous chunk samd_hip.parallel.shard_bounds gives it (the reference's Ray
chunks, eval_vicuna.py:50-65) and rank 0 merges and sorts the per-rank files."""
import json
import os
import time
import uuid
from typing import Callable, List, Optional

import numpy as np
import torch

from .templates import get_conversation_template


def load_questions(question_file: str, begin: Optional[int] = None, end: Optional[int] = None):
    with open(question_file, "r") as f:
        questions = [json.loads(line) for line in f if line.strip()]
    return questions[begin:end]


def reorg_answer_file(answer_file: str):
    """sort by question id and de-duplicate (eval_vicuna.py:247-258)"""
    answers = {}
    with open(answer_file, "r") as fin:
        for line in fin:
            answers[json.loads(line)["question_id"]] = line
    with open(answer_file, "w") as fout:
        for qid in sorted(answers):
            fout.write(answers[qid])


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _clean(output_ids, conv, tokenizer):
    if conv.stop_token_ids:
        hits = [i for i, t in enumerate(output_ids) if t in conv.stop_token_ids]
        if hits:
            output_ids = output_ids[:hits[0]]
    text = tokenizer.decode(output_ids, spaces_between_special_tokens=False)
    if conv.stop_str and text.find(conv.stop_str) > 0:
        text = text[:text.find(conv.stop_str)]
    for special in getattr(tokenizer, "special_tokens_map", {}).values():
        for tok in (special if isinstance(special, list) else [special]):
            text = text.replace(tok, "")
    return text.strip()


def get_model_answers(model, tokenizer, forward_func: Callable, model_id: str, questions: List[dict], answer_file: str,
                      max_new_tokens: int, num_choices: int = 1, template: str = "vicuna", warmup: int = 3, device: str = "cuda",
                      **kwargs):
    if hasattr(model, "eval"):
        model.eval()

    def one_turn(conv, question_text):
        conv.append_message(conv.roles[0], question_text)
        conv.append_message(conv.roles[1], None)
        if conv.stop_str is None:
            conv.stop_str = "</s>"
        inputs = tokenizer([conv.get_prompt()], return_tensors="pt")
        if hasattr(inputs, "to"):
            inputs = inputs.to(device)
        n_in = len(inputs.input_ids[0])
        _sync()
        t0 = time.time()
        output_ids, new_token, step, accept_lengths = forward_func(inputs, model, tokenizer, max_new_tokens, **kwargs)
        _sync()
        return list(output_ids[0][n_in:]), int(new_token), int(step), list(accept_lengths), time.time() - t0

    for _ in range(warmup if questions else 0):
        torch.manual_seed(0)
        conv = get_conversation_template(template)
        for q in questions[0]["turns"]:
            ids, *_ = one_turn(conv, q)
            conv.messages[-1][-1] = _clean(ids, conv, tokenizer)

    all_accept = []
    os.makedirs(os.path.dirname(os.path.abspath(answer_file)), exist_ok=True)
    for question in questions:
        choices = []
        for i in range(num_choices):
            torch.manual_seed(i)
            conv = get_conversation_template(template)
            turns, steps, new_tokens, wall_time, accept = [], [], [], [], []
            for q in question["turns"]:
                try:
                    ids, new_token, step, acc, dt = one_turn(conv, q)
                    output = _clean(ids, conv, tokenizer)
                    all_accept.extend(acc)
                except RuntimeError as e:                          # eval_vicuna.py:218-220
                    print("ERROR question ID: ", question["question_id"], e)
                    output, new_token, step, acc, dt = "ERROR", 0, 0, [], 0.0
                turns.append(output); steps.append(step); new_tokens.append(new_token); wall_time.append(dt); accept.extend(acc)
                conv.messages[-1][-1] = output
            choices.append({"index": i, "turns": turns, "decoding_steps": steps, "new_tokens": new_tokens, "wall_time": wall_time,
                            "accept_lengths": accept})
        with open(answer_file, "a") as fout:
            fout.write(json.dumps({"question_id": question["question_id"], "category": question.get("category", ""),
                                   "answer_id": uuid.uuid4().hex[:22], "model_id": model_id, "choices": choices,
                                   "tstamp": time.time()}) + "\n")
    if all_accept:
        print("#Mean accepted tokens: ", np.mean(all_accept))
    return all_accept


def run_eval(model, tokenizer, forward_func: Callable, model_id: str, question_file: str, question_begin: Optional[int],
             question_end: Optional[int], answer_file: str, max_new_tokens: int, num_choices: int = 1, template: str = "vicuna",
             **kwargs):
    """eval_vicuna.py:20-69 with torch.distributed ranks in place of Ray actors: every rank answers a contiguous chunk of the
    questions (parallel.shard_bounds = the reference's chunking), then the answer records and the accept lengths of all ranks are
    ALL-GATHERED over the process group (RCCL on GPUs: the only collective of the whole evaluation -- BASELINE.json's north star) and
    rank 0 writes the one answer file.  Every rank returns the accept lengths of ALL questions, in question order."""
    import torch.distributed as dist
    from samd_hip.parallel import gather_results, shard_bounds
    questions = load_questions(question_file, question_begin, question_end)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    lo, hi = shard_bounds(len(questions), world, rank)
    part = answer_file if world == 1 else f"{answer_file}.rank{rank}"       # a rank appends as it goes, like the reference's workers
    if world > 1 and os.path.exists(part):
        os.remove(part)
    accept = get_model_answers(model, tokenizer, forward_func, model_id, questions[lo:hi], part, max_new_tokens, num_choices,
                               template=template, **kwargs)
    if distributed:
        lines = []
        if os.path.exists(part):
            with open(part) as fin:
                lines = [ln.rstrip("\n") for ln in fin if ln.strip()]
        records = gather_results([list(ln.encode("utf-8")) for ln in lines])      # per rank: its records as byte rows
        accept_all = gather_results([list(accept)])
        accept = [a for per_rank in accept_all for row in per_rank for a in row]
        if rank == 0:
            with open(answer_file, "a") as fout:
                for per_rank in records:
                    for row in per_rank:
                        fout.write(bytes(row).decode("utf-8") + "\n")
        if os.path.exists(part):
            os.remove(part)
        dist.barrier()
    if rank == 0 and os.path.exists(answer_file):
        reorg_answer_file(answer_file)
    return accept
