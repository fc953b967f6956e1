"""`python -m vptq` - run a VPTQ checkpoint from the command line (prompt completion or a chat loop).

Mirror of the reference's command-line entry (vptq/app_utils.py:17-189, vptq/__main__.py): same
options, same defaults (100 new tokens for a prompt, 500 sampled tokens per chat turn,
pad_token_id 2), same public helpers (`define_basic_args`, `eval_prompt`, `chat_loop`,
`get_chat_loop_generator`, `main`).  The model comes from this package's loader, so every quantised
linear runs on the HIP kernels; there is no hub access in this build, `--model` is a local directory.
"""
from __future__ import annotations

import argparse
import os

PAD_TOKEN_ID = 2   # what the reference passes to generate() for every model
BANNER = "=" * 28 + "chat with the model" + "=" * 28


def define_basic_args() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(
        prog="python -m vptq", formatter_class=argparse.RawTextHelpFormatter,
        description="run a VPTQ-quantised model on MI355X.\n\n"
                    "    python -m vptq --model <checkpoint directory> --prompt \"Explain: ...\"\n"
                    "    python -m vptq --model <checkpoint directory> --chat [--chat-system-prompt \"...\"]\n")
    ap.add_argument("--model", type=str, required=True, help="VPTQ checkpoint (local directory)")
    ap.add_argument("--tokenizer", type=str, default="", help="tokenizer to load (default: the model's)")
    ap.add_argument("--prompt", type=str, default="once upon a time, there ", help="text to continue")
    ap.add_argument("--chat", action="store_true", help="interactive chat instead of one completion")
    ap.add_argument("--chat-system-prompt", type=str, default="you are a math teacher.",
                    help="system message of the chat")
    return ap


def _hub_kwargs() -> dict:
    token = os.getenv("HF_TOKEN")
    return {"token": token} if token is not None else {}


def eval_prompt(model, tokenizer, args):
    """Greedy continuation of `args.prompt`, streamed to stdout (reference app_utils.py:56-62)."""
    import transformers
    batch = tokenizer(args.prompt, return_tensors="pt").to(model.device)
    return model.generate(**batch, streamer=transformers.TextStreamer(tokenizer), max_new_tokens=100,
                          pad_token_id=PAD_TOKEN_ID)


def chat_loop(model, tokenizer, args, read=input):
    """`--chat`: a system message + alternating user / assistant turns through the tokenizer's chat
    template, sampled, until an empty line or `exit` (reference app_utils.py:65-110).  Without
    `--chat`, or with a tokenizer that has no chat template, one prompt completion."""
    import transformers
    if not args.chat:
        return eval_prompt(model, tokenizer, args)
    if getattr(tokenizer, "chat_template", None) is None:
        print("warning: this tokenizer has no chat_template; completing --prompt instead")
        return eval_prompt(model, tokenizer, args)
    print(BANNER)
    print("Press 'exit' to quit")
    history = [{"role": "system", "content": args.chat_system_prompt}]
    streamer = transformers.TextStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True)
    while True:
        text = read("You: ")
        if text in ("", "exit"):
            return history
        history.append({"role": "user", "content": text})
        ids = tokenizer.apply_chat_template(history, add_generation_prompt=True, return_tensors="pt")
        if not hasattr(ids, "shape"):       # newer transformers return a BatchEncoding here
            ids = ids["input_ids"]
        ids = ids.to(model.device)
        print("assistant: ", end="")
        out = model.generate(ids, streamer=streamer, pad_token_id=PAD_TOKEN_ID, max_new_tokens=500,
                             do_sample=True)
        reply = tokenizer.batch_decode(out[:, ids.shape[-1]:], skip_special_tokens=True)[0]
        history.append({"role": "assistant", "content": reply})


def chat_generator(model, tokenizer):
    """The streaming callback a UI drives: `gen(messages, max_tokens, stream=True, temperature=1.0, top_p=1.0)` yields
    the reply piece by piece while `model.generate` runs on a worker thread (reference app_utils.py:127-163)."""
    import threading
    import transformers
    if getattr(tokenizer, "chat_template", None) is None:
        raise Exception("warning: this tokenizer didn't provide chat_template.!!!")   # (the reference's error and wording)

    def chat_loop_generator(messages, max_tokens: int, stream: bool = True, temperature: float = 1.0, top_p: float = 1.0):
        print(BANNER)
        print("Press 'exit' to quit")
        streamer = transformers.TextIteratorStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True)
        enc = tokenizer.apply_chat_template(messages, add_generation_prompt=True, return_tensors="pt", return_dict=True)
        enc = enc.to(model.device)
        worker = threading.Thread(target=model.generate, kwargs=dict(
            enc, streamer=streamer, max_new_tokens=max_tokens, pad_token_id=PAD_TOKEN_ID, do_sample=True,
            temperature=temperature, top_p=top_p))
        worker.start()
        try:
            for piece in streamer:
                yield piece
        finally:
            worker.join()

    return chat_loop_generator


def get_chat_loop_generator(model_id):
    """Load `model_id` (a local checkpoint directory in this build) in fp16 and return its streaming chat callback
    (reference app_utils.py:109-165, what vptq/app.py:17 imports for the Gradio UI)."""
    import transformers
    from vptq_amd.layers.model_base import AutoModelForCausalLM
    model = AutoModelForCausalLM.from_pretrained(model_id, device_map="auto", **_hub_kwargs()).half()
    tokenizer = transformers.AutoTokenizer.from_pretrained(model_id, **_hub_kwargs())
    return chat_generator(model, tokenizer)


def get_valid_args(parser):
    return parser.parse_args()


def main(argv=None):
    import transformers
    from vptq_amd.layers.model_base import AutoModelForCausalLM
    parser = define_basic_args()
    args = parser.parse_args(argv) if argv is not None else get_valid_args(parser)
    print(args)
    model = AutoModelForCausalLM.from_pretrained(args.model, device_map="auto", **_hub_kwargs())
    tokenizer = transformers.AutoTokenizer.from_pretrained(args.tokenizer or args.model, **_hub_kwargs())
    return chat_loop(model, tokenizer, args)
