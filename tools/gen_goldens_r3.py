#!/usr/bin/env python3
"""Round-3 golden vectors captured from the REFERENCE (authoring container only; tools/gen_goldens.py and
tools/gen_goldens_r2.py and their fixtures are left untouched).

  g10_hidden_states.npz   ``output_hidden_states=True`` of the reference forward (valley/model/valley_model.py:281-282,
                          324-330 -> HF LlamaModel: the embeddings, every decoder layer's output, the LAST entry after the
                          final RMSNorm) on the golden model and the "main" prompt (B = 2, left padding, mean pooling):
                          every entry, sub-sampled [:, ::4, ::2], plus the logits of the same call.
Inputs and weights are regenerated from (seed, name, shape) by the tests (valley_amd.weights); the fixture holds outputs only.
Usage: python tools/gen_goldens_r3.py"""
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
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), images=images, attention_mask=torch.from_numpy(mask), output_hidden_states=True,
                    use_cache=False)
    hs = out.hidden_states
    assert len(hs) == G.GCFG["L"] + 1, len(hs)
    d = {f"hs{i}": h.numpy()[:, ::4, ::2].copy() for i, h in enumerate(hs)}
    d["logits"] = out.logits.numpy()[:, ::4].copy()
    d["n"] = np.int64(len(hs))
    np.savez_compressed(os.path.join(GOLD, "g10_hidden_states.npz"), **d)
    print("g10_hidden_states:", len(hs), "entries", hs[0].shape, "max |h|", [round(float(h.abs().max()), 2) for h in hs])


if __name__ == "__main__":
    main()
