"""Corpus -> static suffix automaton (reference: tools/gen_sam_alpaca_sam_only.py:15-49, tools/gen_sam_alpaca.py,
tools/gen_sam_none*.py).

    python -m tools.gen_sam --model_name <tokenizer dir> --sam_data_path dialogues.jsonl --sam_path corpus.sam [--variant samd]

Input: a jsonl of {"prompt": ..., "response": ...} records or a `datasets` directory saved with save_to_disk (what the
reference's prepare_prompts / gen_response pipeline leaves behind).  Each dialogue is tokenised as prompt + response
(:19-35), every vocabulary id is appended as a one-token document so that the root has an edge for every token
(:43-44), all documents go into one automaton with an EOS after each (samd_sam_only/sam/static_sam.py:131-135), and the
result is written as the flat SAMDHIP1 image.  `--sam_data_path none` builds the vocabulary-only automaton of
gen_sam_none*.py.  Construction is the native host builder (a few seconds per 10 M tokens); no GPU is needed."""
import argparse
import json
import os


def load_dialogues(path):
    if path == "none":
        return []
    if os.path.isdir(path):
        from datasets import load_from_disk
        ds = load_from_disk(path)
        return [{"prompt": r["prompt"], "response": r["response"]} for r in ds]
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]


def build_corpus_tokens(dialogues, tokenizer, cutoff_len=None):
    batch = []
    for d in dialogues:
        ids = tokenizer(d["prompt"] + d["response"], padding=False, return_tensors=None)["input_ids"]
        batch.append(list(ids if cutoff_len is None else ids[:cutoff_len]))
    for i in range(len(tokenizer)):
        batch.append([i])
    return batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", required=True)
    ap.add_argument("--sam_data_path", default="none")
    ap.add_argument("--cutoff_len", type=int, default=None)
    ap.add_argument("--sam_path", required=True)
    ap.add_argument("--variant", default="samd_sam_only", choices=["samd_sam_only", "samd"])
    args = ap.parse_args()
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(args.model_name)
    pkg = __import__(args.variant)
    batch = build_corpus_tokens(load_dialogues(args.sam_data_path), tokenizer, args.cutoff_len)
    sam = pkg.build_sam(batch, tokenizer.eos_token_id)
    os.makedirs(os.path.dirname(os.path.abspath(args.sam_path)), exist_ok=True)
    pkg.dump_sam(args.sam_path, sam)
    info = sam._auto.info()
    print(f"{len(batch)} documents -> {info['n_states']} states, {info['n_edges']} edges, {info['device_bytes'] / 1e6:.1f} MB image: {args.sam_path}")


if __name__ == "__main__":
    main()
