"""The tiny configuration the golden fixtures were captured on, and the deterministic inputs.

Shared by tools/gen_goldens.py (authoring container, runs the reference) and the tests (both
boxes).  Shapes are chosen so that the HIP kernels' hard constraints hold: vision width 1024 /
16 heads x 64 (the reference forward hard-codes 1024 and 256 patches, valley_model.py:192),
Llama head_dim 128.
"""
from __future__ import annotations

import numpy as np

from valley_amd import weights as W

GCFG = dict(vocab_text=300, vocab=306, H=256, heads=2, I=512, L=2, eps=1e-5,
            VL=3, VI=256, T=4, seed=11)
PAD = 0
BOS = 1


def special():
    return W.SPECIAL_IDS(GCFG["vocab_text"])


def vision_state():
    return W.clip_vision_weights(GCFG["seed"], hidden=1024, intermediate=GCFG["VI"], layers=GCFG["VL"])


def llama_state():
    c = GCFG
    return W.valley_llama_weights(c["seed"], c["vocab"], c["H"], c["I"], c["L"], std=0.05)


def extra_pool_state(method: str):
    c = GCFG
    H = c["H"]
    s = c["seed"]
    if method == "temporal_importance":
        return {"model.pooling_layer.weight": W.det_normal(s, "pool.w", (1, H * 256), 0.01),
                "model.pooling_layer.bias": W.det_normal(s, "pool.b", (1,), 0.01)}
    if method == "temporal_transformer":
        out = {}
        shapes = {"self_attn.in_proj_weight": (3 * H, H), "self_attn.in_proj_bias": (3 * H,),
                  "self_attn.out_proj.weight": (H, H), "self_attn.out_proj.bias": (H,),
                  "linear1.weight": (2048, H), "linear1.bias": (2048,),
                  "linear2.weight": (H, 2048), "linear2.bias": (H,),
                  "norm1.weight": (H,), "norm1.bias": (H,), "norm2.weight": (H,), "norm2.bias": (H,)}
        for k, shp in shapes.items():
            t = W.det_normal(s, "tde." + k, shp, 0.05)
            if k in ("norm1.weight", "norm2.weight"):
                t = t + 1.0
            out["model.transformer_delta_encoder.layers.0." + k] = t.astype(np.float32)
            out["model.transforemr_adding_layer." + k] = t.astype(np.float32)
        # valley_model.py:89 initialises this to the sinusoid table at training init (zeros at load, :51)
        from oracle.valley_oracle import sinusoid_position_matrix
        out["model.position_matrix"] = sinusoid_position_matrix(2048, H).numpy()
        return out
    return {}


def golden_pixels(n_frames: int, name: str) -> np.ndarray:
    """i.i.d. N(0,1) pixels ~ CLIP-normalised statistics, [n,3,224,224] fp32."""
    return W.det_normal(GCFG["seed"], "px." + name, (n_frames, 3, 224, 224))


def _text(name: str, n: int):
    return W.det_ints(GCFG["seed"], "txt." + name, (n,), 3, GCFG["vocab_text"]).tolist()


def visual_block(T: int, n_patch: int = 256):
    s = special()
    return [s["im_start_token"]] + [s["im_patch_token"]] * n_patch + [s["im_end_token"]] + \
           [s["vi_start_token"]] + [s["vi_frame_token"]] * T + [s["vi_end_token"]]


def _pad_left(seqs):
    S = max(len(x) for x in seqs)
    ids = np.full((len(seqs), S), PAD, np.int64)
    mask = np.zeros((len(seqs), S), np.int64)
    for i, x in enumerate(seqs):
        ids[i, S - len(x):] = x
        mask[i, S - len(x):] = 1
    return ids, mask


def golden_ids(case: str):
    """(input_ids [B,S] int64, attention_mask [B,S] int64) for each fixture case."""
    T = GCFG["T"]
    s = special()
    if case == "main":
        a = [BOS] + _text("m0a", 9) + visual_block(T) + _text("m0b", 12)
        b = [BOS] + _text("m1a", 3) + visual_block(T) + _text("m1b", 7)
        return _pad_left([a, b])
    if case == "mixed":
        b = [BOS] + _text("x1a", 5) + visual_block(T) + _text("x1b", 6)
        a = [BOS] + _text("x0", len(b) - 1)
        return _pad_left([a, b])
    if case == "two_images":
        img = [s["im_start_token"]] + [s["im_patch_token"]] * 256 + [s["im_end_token"]]
        a = [BOS] + _text("t0", 4) + img + _text("t1", 3) + visual_block(T) + _text("t2", 5)
        return _pad_left([a])
    if case == "frame_mismatch":
        a = [BOS] + _text("f0", 4) + visual_block(T + 1) + _text("f1", 5)
        return _pad_left([a])
    if case == "cut":
        a = [BOS] + _text("c0", 4) + visual_block(T, n_patch=255) + _text("c1", 5)
        return _pad_left([a])
    if case == "unbalanced":
        a = [BOS] + _text("u0", 4) + [s["im_start_token"]] + visual_block(T) + _text("u1", 5)
        return _pad_left([a])
    if case == "list":
        a = [BOS] + _text("l0a", 6) + visual_block(2) + _text("l0b", 4)
        b = [BOS] + _text("l1a", 2) + visual_block(3) + _text("l1b", 9)
        return _pad_left([a, b])
    if case == "decode":
        a = [BOS] + _text("d0", 8) + visual_block(T) + _text("d1", 10)
        return _pad_left([a])
    if case == "decode2":
        # chosen (search over 60 seeded prompts with the oracle) so that 8 greedy steps visit 4 different tokens and
        # every top-2 logit gap is > 0.24, ~8x the bf16 path's logit error: the greedy tokens are unambiguous
        a = [BOS] + _text("s30a", 8) + visual_block(T) + _text("s30b", 10)
        return _pad_left([a])
    raise KeyError(case)
