"""Checkpoint reading for real Valley / CLIP weights.

HF layout: ``config.json`` + ``*.safetensors`` (or ``pytorch_model*.bin``) shards.  Keys follow
the reference's contract (valley/model/apply_delta.py:25,30): ``model.embed_tokens.weight``,
``model.layers.N.*``, ``model.norm.weight``, ``lm_head.weight``, ``model.mm_projector.{weight,bias}``,
``model.vision_tower.[vision_model.]*``.  Tensors are handed to the engines as CPU tensors and
packed to device bf16 there."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Tuple

import torch


def _read_shards(path: str) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f, device="cpu"))
        return sd
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    for f in bins:
        sd.update(torch.load(f, map_location="cpu", weights_only=True))
    return sd


def load_valley_checkpoint(path: str, config_cls) -> Tuple[object, Dict[str, torch.Tensor]]:
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    cfg.pop("architectures", None)
    cfg.pop("model_type", None)
    return config_cls(**cfg), _read_shards(path)


def load_clip_checkpoint(path: str):
    from .vision_tower import VisionConfig
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    vc = cfg.get("vision_config", cfg)
    fields = ("hidden_size", "num_attention_heads", "intermediate_size", "num_hidden_layers", "image_size",
              "patch_size", "layer_norm_eps", "hidden_act")
    sd = _read_shards(path)
    sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in sd.items()
          if not k.startswith(("text_model.", "text_projection", "visual_projection", "logit_scale"))}
    return VisionConfig(**{f: vc[f] for f in fields if f in vc}), sd


def apply_delta(base_sd: Dict[str, torch.Tensor], delta_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """valley/model/apply_delta.py:23-33: target = delta + base for base keys; mm_projector / vision
    keys are taken from the delta as is; the embedding rows added for the new tokens stay delta-only."""
    out = {}
    for name, p in delta_sd.items():
        if name not in base_sd:
            out[name] = p
            continue
        b = base_sd[name]
        if p.shape == b.shape:
            out[name] = p + b
        else:
            q = p.clone()
            q[:b.shape[0], :b.shape[1]] += b
            out[name] = q
    return out
