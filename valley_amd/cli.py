"""Command-line front end of the MI355X path.

Mirrors the surface of the reference's ``valley/inference/run_valley.py`` (``init_vision_token`` :13-18, ``main`` :20-57,
the five command-line options :59-66) so that scripts written against it keep working; ``valley/inference/run_valley.py``
in this repository re-exports the two functions.  LoRA merging (reference :26-37) needs ``peft`` and is outside the hot
path: merged checkpoints load through ``ValleyLlamaForCausalLM.from_pretrained``."""
from __future__ import annotations

import argparse
import os
from typing import Optional, Sequence

import torch

from .valley_model import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_VI_END_TOKEN,
                           DEFAULT_VI_START_TOKEN, DEFAULT_VIDEO_FRAME_TOKEN, ValleyLlamaForCausalLM)

# attribute on vision_tower.config  <-  special token whose id it carries (the splice reads these six, splice.py)
TOKEN_BINDINGS = (
    ("im_patch_token", DEFAULT_IMAGE_PATCH_TOKEN),
    ("im_start_token", DEFAULT_IM_START_TOKEN),
    ("im_end_token", DEFAULT_IM_END_TOKEN),
    ("vi_frame_token", DEFAULT_VIDEO_FRAME_TOKEN),
    ("vi_start_token", DEFAULT_VI_START_TOKEN),
    ("vi_end_token", DEFAULT_VI_END_TOKEN),
)

# the reference's default system turn (run_valley.py:46), needed verbatim for identical prompts
SYSTEM_TURN = " ".join((
    "You are Valley, a large language and vision assistant trained by ByteDance.",
    "You are able to understand the visual content or video that the user provides, and assist the user with a variety of tasks using natural language.",
    "Follow the instructions carefully and explain your answers in detail.",
))

GREEDY = {"do_sample": False, "temperature": 0.2, "max_new_tokens": 1024}


def init_vision_token(model, tokenizer) -> None:
    """Write the ids of the six visual special tokens onto the tower's config, where the splice looks them up."""
    cfg = model.get_model().vision_tower.config
    ids = tokenizer.convert_tokens_to_ids([tok for _, tok in TOKEN_BINDINGS])
    for (attr, _), tid in zip(TOKEN_BINDINGS, ids):
        setattr(cfg, attr, int(tid))


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("the MI355X path needs a GPU; there is no CPU execution path")
    return torch.device("cuda")


def load(model_name: str):
    """-> (model on the GPU in eval mode with its token ids bound, tokenizer)."""
    path = os.path.expanduser(model_name)
    if "lora" in path:
        raise NotImplementedError("merge the LoRA adapter offline (peft) and pass the merged checkpoint")
    from transformers import AutoTokenizer
    device = _require_gpu()
    tokenizer = AutoTokenizer.from_pretrained(path)
    model = ValleyLlamaForCausalLM.from_pretrained(path, torch_dtype=torch.bfloat16)
    init_vision_token(model, tokenizer)
    return model.to(device).eval(), tokenizer


def main(args) -> str:
    """One video, one question, one printed answer.  ``args`` carries model_name, query, video_file, system_prompt
    (and vision_tower, unused: the tower comes with the checkpoint)."""
    model, tokenizer = load(args.model_name)
    turns = [{"role": "system", "content": args.system_prompt or SYSTEM_TURN}, {"role": "user", "content": args.query}]
    answer = model.completion(tokenizer, args.video_file, turns, dict(GREEDY), _require_gpu())
    print(answer)
    return answer


def parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, default in (("--model-name", "../../checkpoints/stable-valley-13b-v1"),
                          ("--query", "Describe this video concisely.\n<video>"),
                          ("--video-file", "valley/serve/examples/videos/dc52388394cc9f692d16a95d9833ca07.mp4"),
                          ("--vision-tower", None),
                          ("--system-prompt", "")):
        ap.add_argument(flag, type=str, default=default)
    return ap.parse_args(argv)
