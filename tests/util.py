"""helpers shared by the parity tests (numpy only; no oracle / product imports at module level)."""
import numpy as np


def markov_stream(rng, n, vocab=200, succ=3, noise=0.05):
    table = {}
    out = [int(rng.integers(3, vocab)), int(rng.integers(3, vocab))]
    w = np.array([1.0 / (i + 1) for i in range(succ)])
    w = w / w.sum()
    while len(out) < n:
        key = (out[-2], out[-1])
        if key not in table:
            table[key] = rng.integers(3, vocab, succ).tolist()
        if rng.random() < noise:
            out.append(int(rng.integers(3, vocab)))
        else:
            out.append(int(table[key][rng.choice(succ, p=w)]))
    return out


def split_edges(tab):
    """export dict -> per-state list of (tok,dst) in stored order"""
    out, k = [], 0
    for d in tab["deg"]:
        out.append(list(zip(tab["edge_tok"][k:k + d].tolist(), tab["edge_dst"][k:k + d].tolist())))
        k += d
    return out


def random_parents(rng, n, shape):
    anc = [-1]
    for i in range(1, n):
        if shape == "chain":
            anc.append(i - 1)
        elif shape == "star":
            anc.append(0)
        elif shape == "bushy":
            anc.append(int(rng.integers(max(0, i - 4), i)))
        else:
            anc.append(int(rng.integers(0, i)))
    return anc


def toy_tokenizer_dir(path, n_words=120):
    """a word-level `transformers` tokenizer saved under `path` (what tools.gen_sam's --model_name points at); returns the words"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    words = [f"w{i}" for i in range(n_words)]
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in words:
        vocab[w] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(path)
    return words


def toy_dialogues(rng, words, n=50):
    """{"prompt", "response"} records over the toy vocabulary with shared phrases (so that the automaton has real structure)"""
    phrases = [" ".join(words[int(j)] for j in rng.integers(0, len(words), int(rng.integers(3, 9)))) for _ in range(12)]
    out = []
    for _ in range(n):
        pick = lambda k: " ".join(phrases[int(j)] for j in rng.integers(0, len(phrases), k))
        out.append({"prompt": pick(int(rng.integers(1, 4))) + " ", "response": pick(int(rng.integers(1, 5)))})
    return out


def walk_visited(export, toks_tb):
    """states the reference's transfer_state examines (samd_sam_only/sam/static_sam.py:98-107: every state of the climb, the one where
    it ends included) over all streams of a time-major token matrix, cursors starting at the root; `export` = oracle export().
    Negative tokens count 1 (no state has such an edge: the kernels do not climb for them)."""
    link, deg = export["link"].tolist(), export["deg"].tolist()
    et, ed = export["edge_tok"].tolist(), export["edge_dst"].tolist()
    nxt, k = [], 0
    for d in deg:
        nxt.append(dict(zip(et[k:k + d], ed[k:k + d])))
        k += d
    T, B = len(toks_tb), len(toks_tb[0])
    total = 0
    for b in range(B):
        idx = 0
        for t in range(T):
            tok = int(toks_tb[t][b])
            if tok < 0:
                idx, total = 0, total + 1
                continue
            total += 1
            while idx != 0 and tok not in nxt[idx]:
                idx = link[idx]
                total += 1
            idx = nxt[idx].get(tok, 0)
    return total
