"""Spec-Bench drivers for the MI355X path -- the counterpart of the reference's evaluation/ package
(inference_sam_only.py, inference_samd.py, eval_vicuna.py, eval_llama3.py, speed.py, equal.py) without FastChat, Ray
or shortuuid: conversation templates are spelled out, request-level parallelism is one process per GPU over
torch.distributed (RCCL), results meet in one jsonl."""
from .harness import load_questions, reorg_answer_file, run_eval  # noqa: F401
from .metrics import equal, speed  # noqa: F401
from .templates import get_conversation_template  # noqa: F401
