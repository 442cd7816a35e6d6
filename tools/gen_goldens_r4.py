#!/usr/bin/env python3
"""Round-4 golden vectors captured from the REFERENCE (authoring container only; earlier generators and fixtures untouched).

  g11_attentions.npz   ``output_attentions=True`` of the reference forward (valley/model/valley_model.py:281, 324-330 -> HF
                       LlamaModel's all_self_attns: per decoder layer the softmax probabilities [B, heads, S, S]) on the golden
                       model and the "main" prompt (B = 2, left padding, mean pooling): every layer, query rows sub-sampled
                       [:, :, ::4, :], plus the logits of the same call.
Inputs and weights are regenerated from (seed, name, shape) by the tests (valley_amd.weights); the fixture holds outputs only.
Usage: python tools/gen_goldens_r4.py"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_cfg as G  # noqa: E402
from tools.gen_goldens import build_reference, import_reference  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    torch.manual_seed(0)
    vm = import_reference()
    model = build_reference(vm, "mean")
    try:
        model.config._attn_implementation = "eager"          # HF only materialises the probabilities on the eager path
        model.model.config._attn_implementation = "eager"
    except Exception as e:  # noqa: BLE001
        print("could not select eager attention:", e)
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), images=images, attention_mask=torch.from_numpy(mask), output_attentions=True,
                    use_cache=False)
    at = out.attentions
    assert at is not None and len(at) == G.GCFG["L"], (None if at is None else len(at))
    d = {f"attn{i}": a.numpy()[:, :, ::4].astype(np.float32).copy() for i, a in enumerate(at)}
    d["logits"] = out.logits.numpy()[:, ::4].copy()
    d["n"] = np.int64(len(at))
    np.savez_compressed(os.path.join(GOLD, "g11_attentions.npz"), **d)
    print("g11_attentions:", len(at), "layers", tuple(at[0].shape), "row sums", float(at[0].sum(-1).min()), float(at[0].sum(-1).max()))


if __name__ == "__main__":
    main()
