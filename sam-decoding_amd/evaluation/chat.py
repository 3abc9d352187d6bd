"""Terminal chat over SamdModel.stream_generate -- what the reference's samd_sam_only/inference/cli.py:30-206 (and its
cli_baseline.py / samd/inference/cli.py twins) do with FastChat's ChatIO and conversation classes, without FastChat.

`ChatSession` owns the conversation (evaluation/templates.py) and the commands of the reference loop: `!!exit`, `!!reset`,
`!!remove` (drop the last exchange), `!!regen` (answer the last user message again), `!!save <file>` / `!!load <file>`
(JSON of the message list).  `chat_loop` wires it to an input function, an output function and a SamdModel + tokenizer;
both are injectable, so the loop is covered on CPU with a scripted model (tests/test_harness_cpu.py)."""
import json
import os
from typing import Callable, Iterable, List, Optional

from .templates import Conversation, get_conversation_template


class ChatSession:
    def __init__(self, template: str, system_msg: Optional[str] = None, keep_history: bool = True):
        self.template, self.system_msg, self.keep_history = template, system_msg, keep_history
        self.conv = self._new()

    def _new(self) -> Conversation:
        conv = get_conversation_template(self.template)
        if self.system_msg is not None:
            conv.system = self.system_msg
        return conv

    def reset(self):
        self.conv = self._new()

    # ---- commands ----------------------------------------------------------------------------------------------------
    def remove_last(self) -> bool:
        """drop the last (user, assistant) exchange; False when there is none"""
        if len(self.conv.messages) < 2:
            return False
        del self.conv.messages[-2:]
        return True

    def pop_for_regen(self) -> Optional[str]:
        """remove the last answer and return the user message it answered"""
        m = self.conv.messages
        if len(m) >= 2 and m[-1][0] == self.conv.roles[1] and m[-2][0] == self.conv.roles[0]:
            user = m[-2][1]
            del m[-2:]
            return user
        return None

    def save(self, path: str):
        with open(path, "w") as f:
            json.dump({"template": self.conv.name, "system": getattr(self.conv, "system", ""), "messages": self.conv.messages}, f)

    def load(self, path: str):
        with open(path, "r") as f:
            d = json.load(f)
        self.conv = self._new()
        if d.get("system"):
            self.conv.system = d["system"]
        self.conv.messages = [list(m) for m in d["messages"]]

    # ---- one turn --------------------------------------------------------------------------------------------------------
    def open_turn(self, user_text: str) -> str:
        """append the user message and the empty assistant slot; returns the prompt to tokenize"""
        if not self.keep_history:
            self.reset()
        self.conv.append_message(self.conv.roles[0], user_text)
        self.conv.append_message(self.conv.roles[1], None)
        return self.conv.get_prompt()

    def close_turn(self, answer: str):
        self.conv.messages[-1][1] = answer.strip()


def stream_answer(samd_model, tokenizer, prompt: str, generation_config, emit: Callable[[str], None], stop_str: Optional[str] = None,
                  baseline: bool = False) -> str:
    """tokenize, stream (printing only the new suffix each step, as FastChat's SimpleChatIO does), return the full answer"""
    import torch
    ids = torch.as_tensor([tokenizer(prompt).input_ids if callable(tokenizer) else tokenizer.encode(prompt)])
    gen = samd_model.stream_generate_baseline if baseline and hasattr(samd_model, "stream_generate_baseline") else samd_model.stream_generate
    shown, text = 0, ""
    for chunk in gen(ids, tokenizer, generation_config):
        text = chunk["text"]
        cut = text.find(stop_str) if stop_str else -1
        if cut >= 0:
            text = text[:cut]
        # print whole words only: the tail may still change while a multi-byte piece is being decoded
        stable = text.rsplit(" ", 1)[0] if " " in text and cut < 0 else text
        if len(stable) > shown:
            emit(stable[shown:])
            shown = len(stable)
        if cut >= 0:
            break
    if len(text) > shown:
        emit(text[shown:])
    emit("\n")
    return text


def chat_loop(session: ChatSession, answer: Callable[[str], str], read: Callable[[str], Optional[str]], write: Callable[[str], None],
              debug: bool = False):
    """the reference's REPL (cli.py:76-203).  `answer(prompt) -> text` produces (and streams) one reply."""
    roles = session.conv.roles
    while True:
        try:
            inp = read(f"{roles[0]}: ")
        except EOFError:
            inp = None
        if inp is None or inp == "!!exit" or not inp:
            write("exit...\n")
            return
        if inp == "!!reset":
            write("resetting...\n")
            session.reset()
            continue
        if inp == "!!remove":
            write("removing last message...\n" if session.remove_last() else "No messages to remove.\n")
            continue
        if inp == "!!regen":
            user = session.pop_for_regen()
            if user is None:
                write("No user message to regenerate from.\n")
                continue
            write("regenerating last message...\n")
            inp = user
        elif inp.startswith("!!save"):
            args = inp.split(" ", 1)
            if len(args) != 2:
                write("usage: !!save <filename>\n")
                continue
            name = args[1] if "." in args[1] else args[1] + ".json"
            write(f"saving... {name}\n")
            session.save(name)
            continue
        elif inp.startswith("!!load"):
            args = inp.split(" ", 1)
            if len(args) != 2:
                write("usage: !!load <filename>\n")
                continue
            name = args[1]
            if not os.path.exists(name):
                if not name.endswith(".json") and os.path.exists(name + ".json"):
                    name += ".json"
                else:
                    write(f"file not found: {name}\n")
                    continue
            write(f"loading... {name}\n")
            session.load(name)
            for role, msg in session.conv.messages:
                write(f"{role}: {msg}\n")
            continue
        prompt = session.open_turn(inp)
        if debug:
            write("\n" + repr({"prompt": prompt}) + "\n")
        write(f"{roles[1]}: ")
        try:
            text = answer(prompt)
        except KeyboardInterrupt:
            write("stopped generation.\n")
            session.conv.messages[-1][1] = ""           # keep the turn structure; the partial answer is dropped
            continue
        session.close_turn(text)


def add_common_arguments(parser):
    """the flags of the reference CLIs (cli.py:210-247).  --style / --multiline / --mouse are accepted for command-line
    compatibility; the display is always the plain streaming one."""
    parser.add_argument("--model", type=str, required=True, help="Model name or path.")
    parser.add_argument("--conv-template", type=str, default=None, help="Conversation prompt template.")
    parser.add_argument("--conv-system-msg", type=str, default=None, help="Conversation system message.")
    parser.add_argument("--temperature", type=float, default=0.0)
    parser.add_argument("--max-steps", type=int, default=512)
    parser.add_argument("--no-history", action="store_true")
    parser.add_argument("--style", type=str, default="simple", choices=["simple", "rich", "programmatic"], help="Display style.")
    parser.add_argument("--multiline", action="store_true", help="Enable multiline input (end with an empty line).")
    parser.add_argument("--mouse", action="store_true")
    parser.add_argument("--debug", action="store_true", help="Print useful debug information (e.g., prompts)")
    parser.add_argument("--sam_path", type=str, default=None)
    return parser


def console_reader(multiline: bool) -> Callable[[str], Optional[str]]:
    def read(prompt: str) -> Optional[str]:
        if not multiline:
            return input(prompt)
        lines: List[str] = []
        line = input(prompt + "[multiline, empty line ends]\n")
        while line != "":
            lines.append(line)
            try:
                line = input()
            except EOFError:
                break
        return "\n".join(lines)
    return read


def run_console(args, build_model: Callable[[object, object], object], baseline: bool = False):
    """load the HF model + tokenizer, build the SamdModel through `build_model(lm, tokenizer)`, run the REPL"""
    import sys
    import torch
    from transformers import AutoTokenizer, LlamaForCausalLM
    from samd_sam_only import SamdGenerationConfig
    lm = LlamaForCausalLM.from_pretrained(args.model, torch_dtype=torch.float16, low_cpu_mem_usage=True, device_map="cuda")
    tokenizer = AutoTokenizer.from_pretrained(args.model)
    samd_model = build_model(lm, tokenizer)
    session = ChatSession(args.conv_template or args.model, args.conv_system_msg, keep_history=not args.no_history)
    gcfg = SamdGenerationConfig(max_steps=args.max_steps) if args.temperature < 1e-5 else \
        SamdGenerationConfig(max_steps=args.max_steps, greedy=False, temperature=args.temperature)
    write = lambda s: (sys.stdout.write(s), sys.stdout.flush())
    answer = lambda prompt: stream_answer(samd_model, tokenizer, prompt, gcfg, write, session.conv.stop_str, baseline=baseline)
    try:
        chat_loop(session, answer, console_reader(args.multiline), write, debug=args.debug)
    except KeyboardInterrupt:
        write("exit...\n")
