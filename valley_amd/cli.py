"""Command-line front end of the MI355X path.

Mirrors the surface of the reference's ``valley/inference/run_valley.py`` (``init_vision_token`` :13-18, ``main`` :20-57,
the five command-line options :59-66) so that scripts written against it keep working; ``valley/inference/run_valley.py``
in this repository re-exports the two functions.  A model name containing "lora" takes the reference's adapter branch
(:26-37): the base checkpoint is read, the adapter is merged into its weights on the host
(``valley_amd.checkpoint.merge_lora`` — the arithmetic of peft's ``merge_and_unload``; ``peft`` itself is not needed)
and the merged state goes to the HIP engines.  ``main_v2`` / ``conv_*`` mirror ``run_valley_llamma_v2.py`` and
``run_valley_conv.py``."""
from __future__ import annotations

import argparse
import os
from typing import Optional, Sequence

import torch

from . import runtime
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



def entry_dtype():
    """The dtype the reference's entry points pass to ``from_pretrained`` is ``torch.float16`` (run_valley.py:39,
    run_valley_llamma_v2.py / run_valley_conv.py alike): the same here — it selects libvalley_hip_f16.so — unless
    VALLEY_PRECISION or an earlier model already bound this process to a storage type, which then wins
    (VALLEY_PRECISION=fp32 gives torch.float32: the fp32 validation mode survives the entry points)."""
    if runtime.PRECISION == "fp32":                          # the validation engines (valley_amd/precise.py): never narrowed by an entry point
        return torch.float32
    return runtime.HALF if runtime.half_bound() else torch.float16


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


def load_lora(path: str, device):
    """run_valley.py:26-37: the adapter directory holds ``adapter_config.json``; the base model is the directory itself
    when it also holds a ``config.json``, else ``base_model_name_or_path``; the tokenizer always comes from the base path;
    padding side left."""
    from transformers import AutoTokenizer

    from .checkpoint import load_valley_checkpoint, merge_lora, read_lora_adapter
    from .valley_model import ValleyConfig
    acfg, asd = read_lora_adapter(path)
    base = path if "config.json" in os.listdir(path) else acfg["base_model_name_or_path"]
    config, sd = load_valley_checkpoint(base, ValleyConfig)
    model = ValleyLlamaForCausalLM.from_state_dict(config, merge_lora(sd, acfg, asd), device=device)
    tokenizer = AutoTokenizer.from_pretrained(acfg["base_model_name_or_path"])
    tokenizer.padding_side = "left"
    return model, tokenizer


def load(model_name: str):
    """-> (model on the GPU in eval mode with its token ids bound, tokenizer)."""
    path = os.path.expanduser(model_name)
    device = _require_gpu()
    if "lora" in path:
        model, tokenizer = load_lora(path, device)
    else:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(path)
        model = ValleyLlamaForCausalLM.from_pretrained(path, torch_dtype=entry_dtype())
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


# ---- valley/inference/run_valley_llamma_v2.py ------------------------------------------------------------------------
VALLEY2_7B = "luoruipu1/Valley2-7b"                     # the reference's ModelPath.Valley2_7b (:18-19)
SAMPLED = {"do_sample": True, "temperature": 0.2, "max_new_tokens": 1024}          # its gen_kwargs (:73-77)


def v2_message(query: Optional[str] = None, system_prompt: Optional[str] = None):
    """The four OpenAI-format turns of run_valley_llamma_v2.py:65-70 (incl. the reference's role spelling)."""
    from .valley_model import DEFAULT_VIDEO_TOKEN
    return [{"role": "system", "content": system_prompt or SYSTEM_TURN},
            {"role": "user", "content": "Hi!"},
            {"role": "assistent", "content": "Hi there! How can I help you today?"},
            {"role": "user", "content": query or f"{DEFAULT_VIDEO_TOKEN} Describe the video concisely."}]


def main_v2(video_file: str, model_path: str = VALLEY2_7B):
    """run_valley_llamma_v2.py as a function: Valley2-7b, the fixed 4-turn message, sampled decoding (T = 0.2)."""
    model, tokenizer = load(model_path)
    return model.completion(tokenizer, video_file, v2_message(), dict(SAMPLED), _require_gpu())


# ---- valley/inference/run_valley_conv.py ---------------------------------------------------------------------------------
def conv_bind_tokens(model, tokenizer) -> int:
    """run_valley_conv.py:117-130: add <im_patch> (and, with ``mm_use_im_start_end``, the start/end pairs), write the
    ids onto the tower's config; returns the image token length (256 for ViT-L/14 at 224)."""
    use_se = getattr(model.config, "mm_use_im_start_end", False)
    tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if use_se:
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    vc = model.get_model().vision_tower.config
    vc.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IMAGE_PATCH_TOKEN])[0]
    vc.use_im_start_end = use_se
    if use_se:
        vc.im_start_token, vc.im_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
        vc.vi_start_token, vc.vi_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_VI_START_TOKEN, DEFAULT_VI_END_TOKEN])
        vc.vi_frame_token = tokenizer.convert_tokens_to_ids(DEFAULT_VIDEO_FRAME_TOKEN)
    return (vc.image_size // vc.patch_size) ** 2


def conv_user_turn(qs: str, n_frames: int, image_token_len: int = 256, use_im_start_end: bool = True) -> str:
    """The first human turn of a conversation carries the visual block (run_valley_conv.py:162-166)."""
    if use_im_start_end:
        return qs + "\n" + DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_PATCH_TOKEN * image_token_len + DEFAULT_IM_END_TOKEN + \
            DEFAULT_VI_START_TOKEN + DEFAULT_VIDEO_FRAME_TOKEN * n_frames + DEFAULT_VI_END_TOKEN
    return qs + "\n" + DEFAULT_IMAGE_PATCH_TOKEN * image_token_len


def assistant_out(model, conv, tokenizer, input_ids, image_tensor) -> str:
    """run_valley_conv.py:54-92: sampled generate (T = 0.2, up to 1024 tokens, stop on '###'), strip the role prefixes,
    cut at the separator, newlines removed."""
    from .video import KeywordsStoppingCriteria
    stopping = KeywordsStoppingCriteria(["###"], tokenizer, input_ids)
    with torch.inference_mode():
        output_ids = model.generate(input_ids, images=image_tensor.unsqueeze(0), do_sample=True, temperature=0.2,
                                    max_new_tokens=1024, stopping_criteria=[stopping])
    n_in = input_ids.shape[1]
    n_diff = (input_ids != output_ids[:, :n_in]).sum().item()
    if n_diff > 0:
        print(f"[Warning] {n_diff} output_ids are not the same as the input_ids")
    return conv_clean(tokenizer.batch_decode(output_ids[:, n_in:], skip_special_tokens=True)[0], conv.sep)


def conv_clean(out: str, sep: str = "###") -> str:
    """The text post-processing of run_valley_conv.py:74-92."""
    while True:
        n = len(out)
        out = out.strip()
        for pattern in ("###", "Assistant:", "Response:", "LLaVA:"):
            if out.startswith(pattern):
                out = out[len(pattern):].strip()
        if len(out) == n:
            break
    if sep not in out:
        out += sep
    return out[:out.index(sep)].strip().replace("\n", "") + "\n"


def conv_inference(args, read_line=input, emit=print):
    """The interactive loop of run_valley_conv.py:94-188 ('change video', 'quit', the first turn carrying the visual
    block, conversation memory across turns).  ``read_line`` / ``emit`` make it scriptable."""
    import random

    from transformers import LlamaTokenizer

    from .conversation import conv_templates
    from .video import load_video
    random.seed(42)
    device = _require_gpu()
    tokenizer = LlamaTokenizer.from_pretrained(args.model_name)
    model = ValleyLlamaForCausalLM.from_pretrained(os.path.expanduser(args.model_name), torch_dtype=entry_dtype()).to(device)
    image_token_len = conv_bind_tokens(model, tokenizer)
    use_se = getattr(model.config, "mm_use_im_start_end", False)
    video_path, conv, image_tensor = "", None, None

    def open_video(path):
        video = load_video(path).permute(1, 0, 2, 3)                  # [T,3,224,224]
        return video.to(device), conv_templates[args.conv_mode].copy()

    while True:
        try:
            if not video_path:
                video_path = read_line("Assistant: please input video path. path: ") or args.video_file
                image_tensor, conv = open_video(video_path)
            qs = read_line("human:     ")
            if qs == "change video":
                video_path = read_line("Assistant: please input video path. path: ")
                image_tensor, conv = open_video(video_path)
                qs = read_line("human:     ")
            if qs == "quit":
                break
            if not conv.has_video:
                qs = conv_user_turn(qs, image_tensor.shape[0], image_token_len, use_se)
                conv.has_video = True
            conv.append_message(conv.roles[0], qs)
            input_ids = torch.as_tensor(tokenizer([conv.get_prompt()]).input_ids).to(device)
            answer = assistant_out(model, conv, tokenizer, input_ids, image_tensor)
            conv.append_message(conv.roles[1], answer)
            emit("Assistant: " + answer.strip() + "\n")
        except Exception as e:  # noqa: BLE001 - the reference prints the error as the assistant's turn and keeps going
            emit("Assistant: " + str(e) + "\n")


def conv_parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="interactive video chat (run_valley_conv.py)")
    ap.add_argument("--model-name", type=str, default="../checkpoints/stable-valley-13b-v1/")
    ap.add_argument("--query", type=str, required=False, default="Describe the following video concisely.")
    ap.add_argument("--video_file", type=str, required=False, default="")
    ap.add_argument("--vision-tower", type=str, default=None)
    ap.add_argument("--conv-mode", type=str, default="v1")
    return ap.parse_args(argv)
