"""`python -m vptq` - run a VPTQ checkpoint from the command line (prompt completion or a chat loop).

Mirror of the reference's command-line entry (vptq/app_utils.py:17-189, vptq/__main__.py): same
options, same defaults (100 new tokens for a prompt, 500 sampled tokens per chat turn,
pad_token_id 2), same public helpers (`define_basic_args`, `eval_prompt`, `chat_loop`,
`main`; the UI callback generator of app_utils.py:114-163 is out of this path's scope).  The model comes from this package's loader, so every quantised
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
