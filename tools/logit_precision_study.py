#!/usr/bin/env python3
"""VERDICT r4 item 4, answered on the host: which 16-bit storage points put the logits outside north_star's 1e-3?

The production engines keep GEMM operands and most activations in 16 bits (bf16, or IEEE fp16 on libvalley_hip_f16.so — the
reference's own inference dtype); accumulation, softmax / norm statistics and the residual stream are fp32.  On the golden
configuration the fp16 library lands 4.2e-3 (max-abs) from the reference's fp32 logits, bf16 2.7e-2.  This script evaluates the CPU
oracle (oracle/valley_oracle.py, pinned to the reference fixtures) with `oracle.rounding(dtype, keep=...)`:
  1. every storage point rounded (what the library does),
  2. ONE point at a time kept in fp32 (what that point contributes),
  3. greedily: keep the point that helps most, then the next, ... until the bound is met,
on the golden configuration (tests/golden_cfg.py: the fixture `test_forward_vs_golden` compares with) and on a deeper / wider random model
(closer to the production error budget: the error grows ~sqrt(depth)).

usage: python tools/logit_precision_study.py [fp16|bf16] [golden|deep]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import valley_oracle as O  # noqa: E402
from tests import golden_cfg as G  # noqa: E402
from valley_amd import weights as W  # noqa: E402

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
which = sys.argv[2] if len(sys.argv) > 2 else "golden"
torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))

if which == "golden":
    c = G.GCFG
    lcfg = O.LlamaCfg(hidden=c["H"], heads=c["heads"], intermediate=c["I"], layers=c["L"], vocab=c["vocab"], eps=c["eps"])
    vcfg = O.VisionCfg(intermediate=c["VI"], layers=c["VL"])
    tok = O.TokenIds(**G.special())
    T = c["T"]
    ids, mask = G.golden_ids("main")
    lw, vw = dict(G.llama_state()), G.vision_state()
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
else:                                                       # 12 ViT layers (full width), 8 decoder layers of H = 1024
    T, H, I, L, V = 4, 1024, 2816, 8, 306
    lcfg = O.LlamaCfg(hidden=H, heads=8, intermediate=I, layers=L, vocab=V, eps=1e-5)
    vcfg = O.VisionCfg(layers=12)
    tok = O.TokenIds(**W.SPECIAL_IDS(300))
    lw = W.valley_llama_weights(5, V, H, I, L, std=0.02)
    vw = W.clip_vision_weights(5, layers=12)
    ids = W.synthetic_prompt(7, T, 300).reshape(1, -1)
    mask = np.ones_like(ids)
    images = torch.from_numpy(W.det_normal(5, "px.deep", (T, 3, 224, 224), 1.0)).view(1, T, 3, 224, 224)
valid = torch.from_numpy(mask.astype(bool))


def logits(keep=None):
    with torch.no_grad():
        if keep is None:
            out = O.valley_forward(torch.from_numpy(ids), images, lw, vw, lcfg, vcfg, tok, attention_mask=torch.from_numpy(mask))[0]
        else:
            with O.rounding(dt, keep=keep):
                out = O.valley_forward(torch.from_numpy(ids), images, lw, vw, lcfg, vcfg, tok, attention_mask=torch.from_numpy(mask))[0]
    return out[valid]


ref = logits()
scale = float(ref.abs().max())


def err(keep):
    return float((logits(keep) - ref).abs().max())


base = err(())
name = "fp16" if dt == torch.float16 else "bf16"
print(f"# {name} storage, {which} configuration; reference logits max |x| = {scale:.3f}; bound 1e-3 (north_star)")
print(f"all storage points rounded:             max-abs {base:.3e}")
print(f"nothing rounded but the weights:        max-abs {err(tuple(s for s in O.ROUND_SITES if s != 'weights')):.3e}")
print(f"everything rounded but the weights:     max-abs {err(('weights',)):.3e}")
single = {}
for s in O.ROUND_SITES:
    single[s] = err((s,))
print("one point at a time kept in fp32 (error left, reduction):")
for s, e in sorted(single.items(), key=lambda kv: kv[1]):
    print(f"   {s:14s} {e:.3e}   {100 * (1 - e / base):+6.1f} %")
keep = []
cur = base
print("greedy: keep the most helpful point, then the next ...")
while cur > 1e-3 and len(keep) < len(O.ROUND_SITES):
    best = min((s for s in O.ROUND_SITES if s not in keep), key=lambda s: err(tuple(keep + [s])))
    keep.append(best)
    cur = err(tuple(keep))
    print(f"   + {best:14s} -> {cur:.3e}")
print("kept in fp32:", keep)
