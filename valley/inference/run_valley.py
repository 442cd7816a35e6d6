"""CLI entry point with the reference's surface (valley/inference/run_valley.py:13-66): bind the six
special-token ids onto vision_tower.config, then ``model.completion``.  LoRA merging (:26-37) needs
``peft`` and is outside the hot path; merged checkpoints load through ``from_pretrained``."""
import argparse
import os

import torch

from valley.model.valley_model import ValleyLlamaForCausalLM
from valley.util.config import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN,
                                DEFAULT_VI_END_TOKEN, DEFAULT_VI_START_TOKEN, DEFAULT_VIDEO_FRAME_TOKEN)
from valley.utils import disable_torch_init

DEFAULT_SYSTEM = ("You are Valley, a large language and vision assistant trained by ByteDance. You are able to understand "
                  "the visual content or video that the user provides, and assist the user with a variety of tasks using "
                  "natural language. Follow the instructions carefully and explain your answers in detail.")


def init_vision_token(model, tokenizer):
    """reference run_valley.py:13-18."""
    vc = model.get_model().vision_tower.config
    vc.im_start_token, vc.im_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
    vc.vi_start_token, vc.vi_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_VI_START_TOKEN, DEFAULT_VI_END_TOKEN])
    vc.vi_frame_token = tokenizer.convert_tokens_to_ids(DEFAULT_VIDEO_FRAME_TOKEN)
    vc.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IMAGE_PATCH_TOKEN])[0]


def main(args):
    """reference run_valley.py:20-57."""
    disable_torch_init()
    if not torch.cuda.is_available():
        raise RuntimeError("the MI355X path needs a GPU; there is no CPU execution path")
    device = torch.device("cuda")
    model_name = os.path.expanduser(args.model_name)
    if "lora" in model_name:
        raise NotImplementedError("merge the LoRA adapter offline (peft) and pass the merged checkpoint")
    from transformers import AutoTokenizer
    model = ValleyLlamaForCausalLM.from_pretrained(model_name, torch_dtype=torch.bfloat16)
    tokenizer = AutoTokenizer.from_pretrained(model_name)
    init_vision_token(model, tokenizer)
    model = model.to(device)
    model.eval()
    message = [{"role": "system", "content": args.system_prompt if args.system_prompt else DEFAULT_SYSTEM},
               {"role": "user", "content": args.query}]
    gen_kwargs = dict(do_sample=False, temperature=0.2, max_new_tokens=1024)
    response = model.completion(tokenizer, args.video_file, message, gen_kwargs, device)
    print(response)
    return response


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--model-name", type=str, default="../../checkpoints/stable-valley-13b-v1")
    parser.add_argument("--query", type=str, default="Describe this video concisely.\n<video>")
    parser.add_argument("--video-file", type=str, default="valley/serve/examples/videos/dc52388394cc9f692d16a95d9833ca07.mp4")
    parser.add_argument("--vision-tower", type=str, default=None)
    parser.add_argument("--system-prompt", type=str, default="")
    main(parser.parse_args())
