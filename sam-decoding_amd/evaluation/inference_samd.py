"""Spec-Bench driver for the full variant: SAM + Token Recycle / EAGLE-2 (reference: evaluation/inference_samd.py:13-37,
:41-222).  Same flags as inference_sam_only plus the tree-model ones (scripts/inference_samd.sh:10-20)."""
import argparse
import os

import torch


def samd_forward(inputs, model, tokenizer, max_new_tokens, **kwargs):
    """inference_samd.py:13-37"""
    from samd import SamdGenerationConfig
    max_cache_len = model.lm.config.max_position_embeddings if hasattr(model.lm, "config") else kwargs.get("max_cache_len", 2048)
    out = model.generate(inputs.input_ids, generation_config=SamdGenerationConfig(max_new_tokens=max_new_tokens, max_cache_len=max_cache_len))
    return out.output_ids, out.decode_tokens, out.decode_steps, out.accepet_length_per_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-id", default="vicuna-7b-v1.3-samd")
    ap.add_argument("--model-type", default="vicuna", choices=["vicuna", "llama3"])
    ap.add_argument("--sam-path", default=None)
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--question-begin", type=int, default=None)
    ap.add_argument("--question-end", type=int, default=None)
    ap.add_argument("--answer-file", required=True)
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--num-choices", type=int, default=1)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--samd-n-predicts", type=int, default=40)
    ap.add_argument("--samd-len-threshold", type=int, default=5)
    ap.add_argument("--samd-len-bias", type=int, default=5)
    ap.add_argument("--tree-method", default="token_recycle", choices=["token_recycle", "eagle2"])
    ap.add_argument("--tree-model-path", default=None)
    ap.add_argument("--tree-path", default=None)
    args = ap.parse_args()

    import torch.distributed as dist
    from transformers import AutoModelForCausalLM, AutoTokenizer
    from samd import DraftModel, SamdConfig, SamdModel, load_sam
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
    cfg = SamdConfig(n_predicts=args.samd_n_predicts, len_threshold=args.samd_len_threshold, len_bias=args.samd_len_bias,
                     tree_method=args.tree_method, tree_model_path=args.tree_model_path, tree_path=args.tree_path)
    sam = load_sam(args.sam_path) if args.sam_path else None
    draft = DraftModel(cfg, sam_static=sam, lm=lm, dtype=dtype, device="cuda")
    model = SamdModel(cfg, lm, draft, tokenizer.eos_token_id, dtype, "cuda")
    run_eval(model, tokenizer, samd_forward, args.model_id, args.question_file, args.question_begin, args.question_end,
             args.answer_file, args.max_new_tokens, args.num_choices, template=args.model_type)


if __name__ == "__main__":
    main()
