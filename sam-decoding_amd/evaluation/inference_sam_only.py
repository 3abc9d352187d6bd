"""Spec-Bench driver for the SAM-only variant (reference: evaluation/inference_sam_only.py:12-33, :37-198).

    python -m evaluation.inference_sam_only --model-path <vicuna-7b-v1.3> --sam-path <corpus.sam> --question-file question.jsonl \\
           --answer-file out.jsonl --samd-max-predicts 60 --samd-alpha 4.0 --samd-len-bias 0
    (N GPUs: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m evaluation.inference_sam_only ...)
"""
import argparse
import os

import torch


def sam_only_forward(inputs, model, tokenizer, max_new_tokens, **kwargs):
    """inference_sam_only.py:12-33 -> (output_ids, new_token, step, accept_length_list)"""
    from samd_sam_only import SamdGenerationConfig
    max_cache_len = model.lm.config.max_position_embeddings if hasattr(model.lm, "config") else kwargs.get("max_cache_len", 2048)
    out = model.generate(inputs.input_ids, generation_config=SamdGenerationConfig(max_new_tokens=max_new_tokens, max_cache_len=max_cache_len))
    return out.output_ids, out.decode_tokens, out.decode_steps, out.accepet_length_per_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-id", default="vicuna-7b-v1.3-samd-sam-only")
    ap.add_argument("--model-type", default="vicuna", choices=["vicuna", "llama3"])
    ap.add_argument("--sam-path", default=None)
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--question-begin", type=int, default=None)
    ap.add_argument("--question-end", type=int, default=None)
    ap.add_argument("--answer-file", required=True)
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--num-choices", type=int, default=1)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--samd-max-predicts", type=int, default=40)
    ap.add_argument("--samd-alpha", type=float, default=4.0)
    ap.add_argument("--samd-K", type=int, default=8)
    ap.add_argument("--samd-len-bias", type=int, default=5)
    ap.add_argument("--cache-type", default="static", choices=["static", "dynamic"])
    args = ap.parse_args()

    import torch.distributed as dist
    from transformers import AutoModelForCausalLM, AutoTokenizer
    from samd_sam_only import DraftModel, SamdConfig, SamdModel, load_sam
    from evaluation import run_eval
    local = int(os.environ.get("LOCAL_RANK", 0))
    import samd_hip
    samd_hip.host_waits_by_spinning(local)                 # opt-in, before the HIP context exists (SAMD_SPIN_WAIT=0 disables)
    torch.cuda.set_device(local)
    if int(os.environ.get("WORLD_SIZE", 1)) > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dtype = getattr(torch, args.dtype)
    lm = AutoModelForCausalLM.from_pretrained(args.model_path, torch_dtype=dtype, low_cpu_mem_usage=True).to("cuda")
    tokenizer = AutoTokenizer.from_pretrained(args.model_path)
    cfg = SamdConfig(max_predicts=args.samd_max_predicts, alpha=args.samd_alpha, K=args.samd_K, len_bias=args.samd_len_bias,
                     cache_type=args.cache_type)
    sam = load_sam(args.sam_path) if args.sam_path else None
    draft = DraftModel(cfg, sam_static=sam, lm=lm, dtype=dtype, device="cuda")
    model = SamdModel(cfg, lm, draft, tokenizer.eos_token_id, dtype, "cuda")
    run_eval(model, tokenizer, sam_only_forward, args.model_id, args.question_file, args.question_begin, args.question_end,
             args.answer_file, args.max_new_tokens, args.num_choices, template=args.model_type)


if __name__ == "__main__":
    main()
