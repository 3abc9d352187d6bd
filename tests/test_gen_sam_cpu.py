"""tools.gen_sam (reference: tools/gen_sam_alpaca_sam_only.py:15-49, tools/gen_sam_none*.py) run as the CLI on a toy corpus: the
image it writes must hold exactly the automaton the oracle builds from the same documents (states, edges in dict order, counts,
top-k order).  Host-side only (the native builder); tests/test_gpu_import.py walks the same image on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from util import split_edges, toy_dialogues, toy_tokenizer_dir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sam-decoding_amd")


def run_cli(tmp, dialogues, variant="samd_sam_only", data=True):
    tok_dir, data_path, sam_path = os.path.join(tmp, "tok"), os.path.join(tmp, "d.jsonl"), os.path.join(tmp, "out", "toy.sam")
    words = toy_tokenizer_dir(tok_dir)
    with open(data_path, "w") as f:
        for d in dialogues(words):
            f.write(json.dumps(d) + "\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-m", "tools.gen_sam", "--model_name", tok_dir, "--sam_data_path", data_path if data else "none",
                        "--sam_path", sam_path, "--variant", variant], cwd=PKG, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "documents ->" in r.stdout
    return tok_dir, data_path, sam_path


def corpus_documents(tok_dir, data_path):
    """the documents the CLI builds from (gen_sam_alpaca_sam_only.py:19-44), re-derived independently of the tool's helpers"""
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(tok_dir)
    docs = []
    if data_path:
        for line in open(data_path):
            d = json.loads(line)
            docs.append(list(tok(d["prompt"] + d["response"])["input_ids"]))
    docs += [[i] for i in range(len(tok))]
    return docs, tok.eos_token_id


@pytest.mark.parametrize("with_data", [True, False])
def test_cli_image_equals_oracle_build(tmp_path, with_data):
    import samd_sam_only as SO
    from oracle import sam_oracle as O
    rng = np.random.default_rng(11)
    tok_dir, data_path, sam_path = run_cli(str(tmp_path), lambda words: toy_dialogues(rng, words), data=with_data)
    docs, eos = corpus_documents(tok_dir, data_path if with_data else None)
    sam = SO.load_sam(sam_path)
    ora = O.StaticSAM.build(docs, eos)
    e, t = sam._auto.export(), ora.export()
    for k in ("link", "length", "aux", "deg"):                          # aux = cnt_endpos
        assert e[k].tolist() == t[k].tolist(), k
    tok8, dst8, n8 = ora.export_topk()
    for i, (a, b) in enumerate(zip(split_edges(e), split_edges(t))):
        assert sorted(a) == sorted(b), i                                # the same transitions ...
        assert a[:min(len(a), 8)] == list(zip(tok8[i, :n8[i]].tolist(), dst8[i, :n8[i]].tolist())), i   # ... stored in top-k order
