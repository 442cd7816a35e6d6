#!/usr/bin/env python3
"""Capture the reference's frame preprocessing (the transform chain of load_video,
valley/util/data_util.py:274-281, built from valley/data/video_transform.py classes) on deterministic
uint8 frames -> tests/golden/g7_preprocess.npz.  Authoring container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gen_goldens import import_reference  # noqa: E402
from valley_amd import weights as W  # noqa: E402

import_reference()
from valley.data import video_transform as vt  # noqa: E402

mean = [0.48145466, 0.4578275, 0.40821073]
std = [0.26862954, 0.26130258, 0.27577711]
chain = [vt.TensorToNumpy(), vt.Resize(256), vt.CenterCrop(224), vt.ClipToTensor(channel_nb=3), vt.Normalize(mean=mean, std=std)]
out = {}
for name, (T, H, Wd) in {"landscape": (2, 360, 480), "portrait": (2, 480, 270), "upscale": (1, 200, 310)}.items():
    frames = W.det_ints(5, "vid." + name, (T, H, Wd, 3), 0, 256).astype(np.uint8)          # decord layout [T,H,W,3]
    video = torch.from_numpy(frames).permute(3, 0, 1, 2)                                     # 3 x T x H x W (:265)
    x = video
    for f in chain:
        x = f(x)
    assert tuple(x.shape) == (3, T, 224, 224) and x.dtype == torch.float32
    out[name] = x.numpy()[:, :, ::2, ::2].copy()
    out[name + "_sum"] = np.float64(x.double().sum().item())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_preprocess.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
