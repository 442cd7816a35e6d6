"""Deterministic, machine-independent weight / input generator.

Every tensor is a pure function of (seed, name, shape): a Philox counter RNG keyed by
crc32(name) and the seed.  The same call reproduces bit-identical float32 data in the authoring
container (where the golden fixtures are captured from the reference, tools/gen_goldens.py) and on
the GPU box (where the HIP path and the oracle are fed the same tensors), so fixtures only need to
store outputs.  This is test/bench infrastructure; real checkpoints come in through
``valley_amd.checkpoint``.

Initialisation scales follow the reference stack: Llama linears/embeddings N(0, 0.02^2)
(HF ``LlamaPreTrainedModel._init_weights``), norms = 1, CLIP per
``transformers/models/clip/modeling_clip.py:412-428`` (class/patch/pos embeddings and the
attention/MLP stds derived from hidden size and depth).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np


def _gen(seed: int, name: str) -> np.random.Generator:
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def det_normal(seed: int, name: str, shape: Tuple[int, ...], std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """float32 N(mean, std^2) tensor that depends only on (seed, name, shape)."""
    g = _gen(seed, name)
    out = g.standard_normal(size=shape, dtype=np.float32)
    if std != 1.0:
        out *= np.float32(std)
    if mean != 0.0:
        out += np.float32(mean)
    return out


def det_ints(seed: int, name: str, shape: Tuple[int, ...], low: int, high: int) -> np.ndarray:
    """int64 uniform integers in [low, high)."""
    return _gen(seed, name).integers(low, high, size=shape, dtype=np.int64)


def clip_vision_weights(seed: int, hidden: int = 1024, intermediate: int = 4096, layers: int = 24,
                        image_size: int = 224, patch: int = 14, prefix: str = "") -> Dict[str, np.ndarray]:
    """State dict (HF 5.x flat CLIPVisionModel key names) of a random ViT tower."""
    w: Dict[str, np.ndarray] = {}
    npos = (image_size // patch) ** 2 + 1
    fstd = 1.0
    w["embeddings.class_embedding"] = det_normal(seed, "v.cls", (hidden,), hidden ** -0.5 * fstd)
    w["embeddings.patch_embedding.weight"] = det_normal(seed, "v.patch", (hidden, 3, patch, patch), 0.02)
    w["embeddings.position_embedding.weight"] = det_normal(seed, "v.pos", (npos, hidden), 0.02)
    # LayerNorm affine params are drawn near (1, 0) rather than exactly (1, 0) so that a kernel
    # that drops gamma/beta fails parity.
    w["pre_layrnorm.weight"] = det_normal(seed, "v.preln.w", (hidden,), 0.05, 1.0)
    w["pre_layrnorm.bias"] = det_normal(seed, "v.preln.b", (hidden,), 0.02)
    in_std = hidden ** -0.5 * (2 * layers) ** -0.5 * fstd
    out_std = hidden ** -0.5 * fstd
    fc_std = (2 * hidden) ** -0.5 * fstd
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            w[p + f"self_attn.{nm}.weight"] = det_normal(seed, f"v.{i}.{nm}.w", (hidden, hidden), max(in_std, 0.02))
            w[p + f"self_attn.{nm}.bias"] = det_normal(seed, f"v.{i}.{nm}.b", (hidden,), 0.02)
        w[p + "self_attn.out_proj.weight"] = det_normal(seed, f"v.{i}.o.w", (hidden, hidden), out_std)
        w[p + "self_attn.out_proj.bias"] = det_normal(seed, f"v.{i}.o.b", (hidden,), 0.02)
        w[p + "layer_norm1.weight"] = det_normal(seed, f"v.{i}.ln1.w", (hidden,), 0.05, 1.0)
        w[p + "layer_norm1.bias"] = det_normal(seed, f"v.{i}.ln1.b", (hidden,), 0.02)
        w[p + "layer_norm2.weight"] = det_normal(seed, f"v.{i}.ln2.w", (hidden,), 0.05, 1.0)
        w[p + "layer_norm2.bias"] = det_normal(seed, f"v.{i}.ln2.b", (hidden,), 0.02)
        w[p + "mlp.fc1.weight"] = det_normal(seed, f"v.{i}.fc1.w", (intermediate, hidden), fc_std)
        w[p + "mlp.fc1.bias"] = det_normal(seed, f"v.{i}.fc1.b", (intermediate,), 0.02)
        w[p + "mlp.fc2.weight"] = det_normal(seed, f"v.{i}.fc2.w", (hidden, intermediate), in_std * 2)
        w[p + "mlp.fc2.bias"] = det_normal(seed, f"v.{i}.fc2.b", (hidden,), 0.02)
    w["post_layernorm.weight"] = np.ones((hidden,), np.float32)
    w["post_layernorm.bias"] = np.zeros((hidden,), np.float32)
    if prefix:
        w = {prefix + k: v for k, v in w.items()}
    return w


def valley_llama_weights(seed: int, vocab: int, hidden: int, intermediate: int, layers: int,
                         mm_hidden: int = 1024, std: float = 0.02) -> Dict[str, np.ndarray]:
    """State dict with the reference's key names (valley/model/apply_delta.py:25-30 names the
    non-base keys ``model.mm_projector.{weight,bias}``)."""
    w: Dict[str, np.ndarray] = {}
    w["model.embed_tokens.weight"] = det_normal(seed, "l.embed", (vocab, hidden), std)
    for i in range(layers):
        p = f"model.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            w[p + f"self_attn.{nm}.weight"] = det_normal(seed, f"l.{i}.{nm}", (hidden, hidden), std)
        w[p + "mlp.gate_proj.weight"] = det_normal(seed, f"l.{i}.gate", (intermediate, hidden), std)
        w[p + "mlp.up_proj.weight"] = det_normal(seed, f"l.{i}.up", (intermediate, hidden), std)
        w[p + "mlp.down_proj.weight"] = det_normal(seed, f"l.{i}.down", (hidden, intermediate), std)
        w[p + "input_layernorm.weight"] = det_normal(seed, f"l.{i}.ln1", (hidden,), 0.05, 1.0)
        w[p + "post_attention_layernorm.weight"] = det_normal(seed, f"l.{i}.ln2", (hidden,), 0.05, 1.0)
    w["model.norm.weight"] = det_normal(seed, "l.norm", (hidden,), 0.05, 1.0)
    w["lm_head.weight"] = det_normal(seed, "l.lm_head", (vocab, hidden), std)
    w["model.mm_projector.weight"] = det_normal(seed, "l.mmproj.w", (hidden, mm_hidden), std)
    w["model.mm_projector.bias"] = det_normal(seed, "l.mmproj.b", (hidden,), std)
    return w


def synthetic_prompt(seed: int, n_frames: int, vocab_text: int = 32000, n_pre: int = 27, n_post: int = 32,
                     bos: int = 1, ids=None) -> np.ndarray:
    """SURVEY.md §8(d) prompt: [BOS] + n_pre text + <im_start> <im_patch>x256 <im_end> <vi_start>
    <vi_frame>xT <vi_end> + n_post text  =>  S = 320 + T for the defaults."""
    ids = ids or SPECIAL_IDS(vocab_text)
    pre = det_ints(seed, "prompt.pre", (n_pre,), 3, vocab_text)
    post = det_ints(seed, "prompt.post", (n_post,), 3, vocab_text)
    seq = [bos] + pre.tolist() + [ids["im_start_token"]] + [ids["im_patch_token"]] * 256 + [ids["im_end_token"]] \
        + [ids["vi_start_token"]] + [ids["vi_frame_token"]] * n_frames + [ids["vi_end_token"]] + post.tolist()
    return np.asarray(seq, dtype=np.int64)


def SPECIAL_IDS(vocab_text: int = 32000):
    """Token-id order of valley_model.py:357-360: patch, frame, im_start, im_end, vi_start, vi_end."""
    return dict(im_patch_token=vocab_text, vi_frame_token=vocab_text + 1, im_start_token=vocab_text + 2,
                im_end_token=vocab_text + 3, vi_start_token=vocab_text + 4, vi_end_token=vocab_text + 5)
