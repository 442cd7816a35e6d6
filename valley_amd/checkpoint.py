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


def _model_shards(path: str, pattern: str):
    """Model shards only: a peft adapter saved into the same directory (run_valley.py:27-29 — a ``config.json`` next to
    ``adapter_config.json`` means the base model IS that directory) must not be read as model weights."""
    return sorted(f for f in glob.glob(os.path.join(path, pattern)) if not os.path.basename(f).startswith("adapter_model"))


def _read_shards(path: str) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    st = _model_shards(path, "*.safetensors")
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f, device="cpu"))
        return sd
    bins = _model_shards(path, "pytorch_model*.bin")
    if not bins:
        raise FileNotFoundError(f"no model *.safetensors or pytorch_model*.bin under {path}")
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


NON_BASE_OK = ("model.mm_projector.weight", "model.mm_projector.bias")
RESIZED_OK = ("model.embed_tokens.weight", "lm_head.weight")


def apply_delta(base_sd: Dict[str, torch.Tensor], delta_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """valley/model/apply_delta.py:23-33 on state dicts: target = delta + base for every key the base has; keys the base
    lacks must be the projector or vision-tower weights (taken from the delta as they are, :25); a shape mismatch is
    allowed only for the two vocabulary-sized matrices, whose first base-vocab rows get the base added (:30-33).
    Same assertion texts as the reference."""
    out = {}
    for name, p in delta_sd.items():
        if name not in base_sd:
            assert name in NON_BASE_OK or "vision_tower" in name, f"{name} not in base model"
            out[name] = p
            continue
        b = base_sd[name]
        if p.shape == b.shape:
            out[name] = p + b
        else:
            assert name in RESIZED_OK, f"{name} dimension mismatch: {p.shape} vs {b.shape}"
            q = p.clone()
            q[:b.shape[0], :b.shape[1]] += b
            out[name] = q
    return out


def make_delta(base_sd: Dict[str, torch.Tensor], target_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of apply_delta (valley/model/make_delta.py:11-31): delta = target - base on the base's keys / rows."""
    out = {}
    for name, p in target_sd.items():
        if name not in base_sd:
            assert name in NON_BASE_OK or "vision_tower" in name, f"{name} not in base model"
            out[name] = p
            continue
        b = base_sd[name]
        q = p.clone()
        if p.shape == b.shape:
            q -= b
        else:                                            # only the two vocabulary-sized matrices may differ (make_delta.py:28)
            assert name in RESIZED_OK, f"{name} dimension mismatch: {p.shape} vs {b.shape}"
            q[:b.shape[0], :b.shape[1]] -= b
        out[name] = q
    return out


def read_lora_adapter(path: str):
    """A peft LoRA adapter directory: ``adapter_config.json`` (r, lora_alpha, fan_in_fan_out, ...) +
    ``adapter_model.safetensors`` / ``adapter_model.bin``."""
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st, device="cpu")
    else:
        sd = torch.load(os.path.join(path, "adapter_model.bin"), map_location="cpu", weights_only=True)
    return cfg, sd


def merge_lora(sd: Dict[str, torch.Tensor], adapter_cfg: dict, adapter_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """What ``PeftModel.from_pretrained(model, path).merge_and_unload()`` does to the weights
    (valley/inference/run_valley.py:26-37; peft's published LoRA merge rule, peft/tuners/lora/layer.py ``get_delta_weight``):
    for every adapted Linear,  W <- W + (lora_alpha / r) * B @ A  (transposed when ``fan_in_fan_out``), computed in fp32.
    Adapter keys look like ``base_model.model.<module path>.lora_A[.<adapter name>].weight``; modules listed in
    ``modules_to_save`` are stored whole and replace the base tensor.  Returns a new state dict."""
    r = int(adapter_cfg["r"])
    scale = float(adapter_cfg["lora_alpha"]) / r
    if adapter_cfg.get("use_rslora"):
        scale = float(adapter_cfg["lora_alpha"]) / (r ** 0.5)
    fifo = bool(adapter_cfg.get("fan_in_fan_out", False))
    out = dict(sd)

    def base_key(k: str, marker: str) -> str:
        mod = k[:k.index(marker)]
        for pre in ("base_model.model.", "base_model."):
            if mod.startswith(pre):
                mod = mod[len(pre):]
                break
        return mod + ".weight"

    def strip(k: str) -> str:
        for pre in ("base_model.model.", "base_model."):
            if k.startswith(pre):
                return k[len(pre):]
        return k

    pairs, unused = {}, []
    for k, v in adapter_sd.items():
        hit = False
        for tag in ("lora_A", "lora_B"):
            m = f".{tag}."
            if m in k and k.endswith("weight"):
                pairs.setdefault(base_key(k, m), {})[tag] = v
                hit = True
        if hit:
            continue
        if ".modules_to_save." in k:                      # in-memory key form: ...embed_tokens.modules_to_save.default.weight
            mod = base_key(k, ".modules_to_save.")[:-len(".weight")]
            out[mod + "." + k.rsplit(".", 1)[1]] = v
        elif strip(k) in out and "lora_" not in k:
            # peft's get_peft_model_state_dict strips ``modules_to_save.<adapter>.`` when it SAVES, so a module saved whole
            # arrives as a plain base key (base_model.model.lm_head.weight): a whole-tensor replacement
            out[strip(k)] = v
        else:
            unused.append(k)
    if unused:
        # lora_embedding_A/B, DoRA magnitudes, bias terms of an unknown module ...: merging around them would give a
        # silently wrong model
        raise KeyError(f"LoRA adapter tensors this merge does not understand: {sorted(unused)[:6]}"
                       + (f" (+{len(unused) - 6} more)" if len(unused) > 6 else ""))
    for wkey, ab in pairs.items():
        if "lora_A" not in ab or "lora_B" not in ab:
            raise KeyError(f"LoRA adapter holds only one factor for {wkey}")
        if wkey not in out:
            raise KeyError(f"LoRA target {wkey} is not in the base checkpoint")
        delta = (ab["lora_B"].float() @ ab["lora_A"].float()) * scale
        if fifo:
            delta = delta.t()
        w = out[wkey]
        if tuple(delta.shape) != tuple(w.shape):
            raise ValueError(f"LoRA delta {tuple(delta.shape)} does not fit {wkey} {tuple(w.shape)}")
        out[wkey] = (w.float() + delta).to(w.dtype)
    return out
