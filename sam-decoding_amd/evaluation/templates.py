"""Conversation templates (the reference takes them from FastChat: eval_vicuna.py:99-105, eval_llama3.py).

`vicuna` is FastChat's "vicuna_v1.1" (system sentence, roles USER/ASSISTANT, separators " " and "</s>"), which is what
`get_conversation_template("vicuna")` resolves to; `llama3` is Meta's Llama-3-Instruct header format."""
from typing import List, Optional


class Conversation:
    name = "base"
    roles = ("USER", "ASSISTANT")
    stop_str: Optional[str] = None
    stop_token_ids: List[int] = []

    def __init__(self):
        self.messages: List[List[Optional[str]]] = []

    def append_message(self, role: str, message: Optional[str]):
        self.messages.append([role, message])

    def get_prompt(self) -> str:
        raise NotImplementedError


class VicunaConversation(Conversation):
    name = "vicuna_v1.1"
    system = ("A chat between a curious user and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the user's questions.")
    sep, sep2 = " ", "</s>"

    def get_prompt(self) -> str:
        seps = [self.sep, self.sep2]
        out = self.system + seps[0]
        for i, (role, message) in enumerate(self.messages):
            out += (role + ": " + message + seps[i % 2]) if message else (role + ":")
        return out


class Llama3Conversation(Conversation):
    name = "llama-3"
    roles = ("user", "assistant")
    system = ""
    stop_str = "<|eot_id|>"

    def get_prompt(self) -> str:
        out = "<|begin_of_text|>"
        if self.system:
            out += "<|start_header_id|>system<|end_header_id|>\n\n" + self.system + "<|eot_id|>"
        for role, message in self.messages:
            out += "<|start_header_id|>" + role + "<|end_header_id|>\n\n"
            if message:
                out += message.strip() + "<|eot_id|>"
        return out


def get_conversation_template(name: str) -> Conversation:
    key = name.lower()
    if "llama-3" in key or "llama3" in key:
        return Llama3Conversation()
    if "vicuna" in key:
        return VicunaConversation()
    raise ValueError(f"no conversation template for '{name}'")
