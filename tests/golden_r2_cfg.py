"""Deterministic inputs of the round-2 fixtures (tools/gen_goldens_r2.py, authoring container) — shared with the tests on
both boxes.  Everything is a pure function of (seed, name, shape) through valley_amd.weights."""
from __future__ import annotations

import numpy as np

from valley_amd import weights as W

SEED = 23
# op-level shapes: Llama pieces at head_dim 128 (2 heads), CLIP pieces at the real 1024 / 16 x 64 geometry
OPS = dict(H=256, heads=2, I=512, S=75, pad=9, VI=256, rope_positions=[0, 1, 327, 2047])


def op_input(name, shape, std=1.0):
    return W.det_normal(SEED, "opx." + name, tuple(shape), std)


def op_weights(name, shape, std=0.05, mean=0.0):
    return W.det_normal(SEED, "opw." + name, tuple(shape), std, mean)


def delta_states():
    """(base state dict, delta state dict, dims): a tiny LLaMA base and a Valley delta with 6 more vocabulary rows, the
    projector and a vision-tower tensor — every branch of valley/model/apply_delta.py:23-33."""
    dims = dict(vocab_base=40, vocab=46, H=64, I=128, L=1, heads=2)
    base = W.valley_llama_weights(SEED, dims["vocab_base"], dims["H"], dims["I"], dims["L"], std=0.05)
    base = {k: v for k, v in base.items() if "mm_projector" not in k}
    delta = W.valley_llama_weights(SEED + 1, dims["vocab"], dims["H"], dims["I"], dims["L"], std=0.05)
    return base, delta, dims
