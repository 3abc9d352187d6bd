"""python -m samd.inference.cli --model <path> [--sam_path <file>] [--tree_method token_recycle|eagle|eagle2]
[--tree_model_path <dir>] -- chat with SAM sequence drafts + an auxiliary tree-draft model (reference: samd/inference/cli.py)."""
import argparse

import torch

from evaluation.chat import add_common_arguments, run_console
from samd import DraftModel, SamdConfig, SamdModel, load_sam


def build(args):
    def build_model(lm, tokenizer):
        sam = load_sam(args.sam_path) if args.sam_path is not None else None
        cfg = SamdConfig(tree_method=args.tree_method, tree_model_path=args.tree_model_path)
        draft = DraftModel(cfg, sam_dyn=None, sam_static=sam, lm=lm, dtype=torch.float16, device="cuda")
        return SamdModel(cfg, lm, draft, tokenizer.eos_token_id, dtype=torch.float16, device="cuda")
    return build_model


def main(argv=None):
    parser = add_common_arguments(argparse.ArgumentParser())
    parser.add_argument("--tree_method", type=str, default="token_recycle")
    parser.add_argument("--tree_model_path", type=str, default=None)
    args = parser.parse_args(argv)
    run_console(args, build(args))


if __name__ == "__main__":
    main()
