"""python -m samd_sam_only.inference.cli_baseline --model <path> -- the same chat with plain autoregressive decoding: the
draft is limited to the start token (max_predicts = 1), as the reference's cli_baseline.py:49 does."""
import argparse

import torch

from evaluation.chat import add_common_arguments, run_console
from samd_sam_only import DraftModel, SamdConfig, SamdModel


def build(args):
    def build_model(lm, tokenizer):
        cfg = SamdConfig(max_predicts=1)
        draft = DraftModel(cfg, sam_dyn=None, sam_static=None, lm=lm, dtype=torch.float16, device="cuda")
        return SamdModel(cfg, lm, draft, tokenizer.eos_token_id, dtype=torch.float16, device="cuda")
    return build_model


def main(argv=None):
    args = add_common_arguments(argparse.ArgumentParser()).parse_args(argv)
    run_console(args, build(args), baseline=True)


if __name__ == "__main__":
    main()
