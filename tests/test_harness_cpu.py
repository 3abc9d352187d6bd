"""The Spec-Bench harness counterpart (sam-decoding_amd/evaluation) on CPU with a toy tokenizer and a toy forward:
prompt templates, answer records (the fields evaluation/speed.py of the reference consumes), warm-up count, stop-string
clean-up, de-duplicating re-org, speed / equal metrics."""
import json

import pytest

torch = pytest.importorskip("torch")


class ToyTokenizer:
    """whitespace tokenizer with a BOS; decode joins with spaces."""
    special_tokens_map = {"eos_token": "</s>", "bos_token": "<s>"}

    def __init__(self):
        self.vocab, self.inv = {"<s>": 0, "</s>": 2}, {0: "<s>", 2: "</s>"}

    def _id(self, w):
        if w not in self.vocab:
            self.vocab[w] = len(self.vocab) + 10
            self.inv[self.vocab[w]] = w
        return self.vocab[w]

    def __call__(self, texts, return_tensors=None):
        single = isinstance(texts, str)
        ids = [[0] + [self._id(w) for w in t.split()] for t in ([texts] if single else texts)]

        class Enc:
            pass
        e = Enc()
        e.input_ids = ids[0] if single else (torch.tensor(ids) if return_tensors == "pt" else ids)
        e.to = lambda device: e
        return e

    def decode(self, ids, **kw):
        return " ".join(self.inv[int(i)] for i in ids)


def test_templates():
    from evaluation import get_conversation_template
    conv = get_conversation_template("vicuna")
    conv.append_message(conv.roles[0], "Hi there")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == ("A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, "
                                 "detailed, and polite answers to the user's questions. USER: Hi there ASSISTANT:")
    conv.messages[-1][-1] = "Hello!"
    conv.append_message(conv.roles[0], "Again")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt().endswith("USER: Hi there ASSISTANT: Hello!</s>USER: Again ASSISTANT:")
    l3 = get_conversation_template("llama3")
    l3.append_message(l3.roles[0], "Q")
    l3.append_message(l3.roles[1], None)
    assert l3.get_prompt() == "<|begin_of_text|><|start_header_id|>user<|end_header_id|>\n\nQ<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"


def test_run_eval_speed_equal(tmp_path):
    from evaluation import equal, run_eval, speed
    tok = ToyTokenizer()
    qfile = tmp_path / "q.jsonl"
    qs = [{"question_id": 3, "category": "qa", "turns": ["what is up", "and then"]},
          {"question_id": 1, "category": "summarization", "turns": ["summarise this text"]}]
    qfile.write_text("".join(json.dumps(q) + "\n" for q in qs))
    calls = []

    def forward(inputs, model, tokenizer, max_new_tokens, accept=(3, 2)):
        calls.append(len(inputs.input_ids[0]))
        new = [tok._id(w) for w in "fine thanks </s> junk".split()]          # the stop string must cut the answer
        return [inputs.input_ids[0].tolist() + new], 5, 2, list(accept)

    ans = tmp_path / "out" / "a.jsonl"
    run_eval(object(), tok, forward, "toy", str(qfile), None, None, str(ans), max_new_tokens=16)
    rows = [json.loads(l) for l in ans.read_text().splitlines()]
    assert [r["question_id"] for r in rows] == [1, 3]                         # re-organised by question id
    assert len(calls) == 3 * 2 + 3                                            # 3 warm-up passes over question 0 (2 turns) + 3 turns
    c = rows[1]["choices"][0]
    assert c["turns"] == ["fine thanks", "fine thanks"] and c["new_tokens"] == [5, 5] and c["decoding_steps"] == [2, 2]
    assert c["accept_lengths"] == [3, 2, 3, 2] and len(c["wall_time"]) == 2 and rows[1]["model_id"] == "toy"
    # baseline file: same answers, slower
    base = tmp_path / "b.jsonl"
    slow = []
    for r in rows:
        r2 = json.loads(json.dumps(r))
        for ch in r2["choices"]:
            ch["wall_time"] = [10.0 for _ in ch["wall_time"]]
        slow.append(r2)
    base.write_text("".join(json.dumps(r) + "\n" for r in slow))
    for r in rows:                                                            # make the method's timing deterministic
        for ch in r["choices"]:
            ch["wall_time"] = [1.0 for _ in ch["wall_time"]]
    ans.write_text("".join(json.dumps(r) + "\n" for r in rows))
    tps, tps0, ratio, acc = speed(str(ans), str(base), lambda text: len(text.split()) + 1, task="overall", report=False)
    assert tps == pytest.approx(5.0) and tps0 == pytest.approx(0.2) and ratio == pytest.approx(25.0) and acc == [3, 2, 3, 2, 3, 2]
    assert speed(str(ans), str(base), lambda text: len(text.split()) + 1, task="qa", report=False)[3] == [3, 2, 3, 2]
    assert equal(str(ans), str(base), report=False)
    slow[0]["choices"][0]["turns"][0] = "different"
    base.write_text("".join(json.dumps(r) + "\n" for r in slow))
    assert not equal(str(ans), str(base), report=False)


def test_gen_sam_tool_builds_corpus_with_vocab_documents(tmp_path):
    """tools.gen_sam: prompt + response per dialogue, then every vocabulary id as a one-token document
    (tools/gen_sam_alpaca_sam_only.py:39-44); the automaton must equal a direct build of that corpus."""
    import samd_sam_only as SO
    from tools.gen_sam import build_corpus_tokens, load_dialogues

    class Tok(ToyTokenizer):
        def __call__(self, text, padding=False, return_tensors=None):
            return {"input_ids": [0] + [self._id(w) for w in text.split()]}

        def __len__(self):
            return 40
    tok = Tok()
    p = tmp_path / "d.jsonl"
    p.write_text(json.dumps({"prompt": "a b c ", "response": "a b d"}) + "\n" + json.dumps({"prompt": "e a ", "response": "b c"}) + "\n")
    batch = build_corpus_tokens(load_dialogues(str(p)), tok)
    assert len(batch) == 2 + 40 and batch[2:5] == [[0], [1], [2]] and len(batch[0]) == 7
    sam = SO.build_sam(batch, 2)
    assert len(sam.states[0].next) >= 40                      # the root has an edge for every vocabulary id
    assert load_dialogues("none") == []


# --------------------------------------------------------------------------------------------------
# chat REPL (evaluation/chat.py; reference samd_sam_only/inference/cli.py:76-203)
# --------------------------------------------------------------------------------------------------
class _EchoModel:
    """stands in for SamdModel.stream_generate: 'answers' with the number of prompt tokens, a few characters per step"""

    def __init__(self):
        self.calls = 0

    def stream_generate(self, input_ids, tokenizer, generation_config=None):
        self.calls += 1
        full = f"answer {self.calls} to {input_ids.shape[-1]} tokens</s>ignored"
        for i in range(3, len(full) + 3, 3):
            yield {"text": full[:i]}


class _WordTokenizer:
    def encode(self, text):
        return list(range(len(text.split())))

    def __call__(self, text):
        class R:
            input_ids = self.encode(text)
        return R


def test_chat_loop_commands(tmp_path):
    from evaluation.chat import ChatSession, chat_loop, stream_answer
    model, tok = _EchoModel(), _WordTokenizer()
    out = []
    save = str(tmp_path / "conv")
    script = iter(["hello there", "second question", "!!regen", "!!save " + save, "!!remove", "!!remove", "!!remove",
                   "!!load " + save, "!!load nowhere", "!!save", "!!reset", "after reset", "!!exit"])
    session = ChatSession("vicuna")
    answer = lambda prompt: stream_answer(model, tok, prompt, None, out.append, stop_str="</s>")
    chat_loop(session, answer, lambda p: next(script), out.append)
    text = "".join(out)
    assert "answer 1 to" in text and "answer 2 to" in text and "regenerating last message" in text and "answer 3 to" in text
    assert "ignored" not in text and "</s>" not in text                 # the stop string cuts the stream
    assert text.count("removing last message") == 2 and "No messages to remove." in text
    assert "file not found: nowhere" in text and "usage: !!save <filename>" in text and text.endswith("exit...\n")
    saved = json.load(open(save + ".json"))
    assert [m[0] for m in saved["messages"]] == ["USER", "ASSISTANT", "USER", "ASSISTANT"]
    assert saved["messages"][3][1].startswith("answer 3")               # the regenerated reply replaced the second one
    # after !!reset only the last exchange is in the conversation, and prompts carry the history while it exists
    assert [m[1] for m in session.conv.messages][0] == "after reset" and len(session.conv.messages) == 2
    assert model.calls == 4


def test_chat_without_history_and_cli_flags():
    import argparse
    from evaluation.chat import ChatSession, add_common_arguments
    s = ChatSession("llama-3-8b", system_msg="be brief", keep_history=False)
    p1 = s.open_turn("one"); s.close_turn("uno")
    p2 = s.open_turn("two")
    assert "one<|eot_id|>" not in p2 and "be brief" in p2 and p2.endswith("<|start_header_id|>assistant<|end_header_id|>\n\n")
    args = add_common_arguments(argparse.ArgumentParser()).parse_args(["--model", "m", "--sam_path", "x.samd", "--max-steps", "7", "--no-history"])
    assert (args.model, args.sam_path, args.max_steps, args.no_history, args.temperature, args.style) == ("m", "x.samd", 7, True, 0.0, "simple")
