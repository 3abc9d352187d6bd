"""python -m samd_sam_only.inference.cli --model <path> [--sam_path <file>] -- chat with SAM-drafted decoding
(reference: samd_sam_only/inference/cli.py; the REPL lives in evaluation/chat.py)."""
import argparse

import torch

from evaluation.chat import add_common_arguments, run_console
from samd_sam_only import DraftModel, SamdConfig, SamdModel, load_sam


def build(args):
    def build_model(lm, tokenizer):
        sam = load_sam(args.sam_path) if args.sam_path is not None else None
        cfg = SamdConfig()
        draft = DraftModel(cfg, sam_dyn=None, sam_static=sam, lm=lm, dtype=torch.float16, device="cuda")
        return SamdModel(cfg, lm, draft, tokenizer.eos_token_id, dtype=torch.float16, device="cuda")
    return build_model


def main(argv=None):
    args = add_common_arguments(argparse.ArgumentParser()).parse_args(argv)
    run_console(args, build(args))


if __name__ == "__main__":
    main()
