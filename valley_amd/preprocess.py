"""GPU frame preprocessing: decoded uint8 frames [T,H,W,3] -> CLIP-normalised bf16/fp32 [T,3,224,224]
(SURVEY.md §8f N2 — the step right before the hot path; at >3 k frames/s/GPU of ViT throughput the
reference's CPU PIL chain, valley/util/data_util.py:262-281, becomes the bottleneck).

Bit-exact with the reference chain up to the final float normalisation: short side to 256 with
Pillow's 8-bit bilinear resample (the 'nearest' default of valley/data/video_transform.py:269 selects
PIL.Image.BILINEAR through the inverted test at :63-66), centre crop 224 (:542-544), /255, CLIP mean/std.
Pillow's algorithm (libImaging/Resample.c) is separable: the coefficient tables are tiny and built here
on the host in float64 exactly as precompute_coeffs()/normalize_coeffs_8bpc() do; the two convolution
passes + crop + normalise run in vly_resize_h_u8 / vly_resize_v_norm."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch

from . import lib as _lib
from . import runtime
from .ops import _chk, _stream

PRECISION_BITS = 22
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resize_sizes(im_h: int, im_w: int, size: int = 256) -> Tuple[int, int]:
    """valley/data/video_transform.py:36-41,74-81 (no resize when the short side already matches)."""
    if (im_w <= im_h and im_w == size) or (im_h <= im_w and im_h == size):
        return im_h, im_w
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)


def resample_tables(in_size: int, out_size: int):
    """(bounds int32 [out,2] = (first input index, tap count), taps int32 [out,ksize]) of Pillow's bilinear
    resample from in_size to out_size, 22-bit fixed point."""
    scale = in_size / out_size
    fscale = scale if scale > 1.0 else 1.0
    support = fscale                                        # bilinear filter support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / fscale
    one = float(1 << PRECISION_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        ws = []
        for x in range(hi - lo):
            a = abs((x + lo - center + 0.5) * inv)
            ws.append(1.0 - a if a < 1.0 else 0.0)
        tot = sum(ws)
        for x, w in enumerate(ws):
            if tot != 0.0:
                w = w / tot
            taps[o, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
        bounds[o] = (lo, hi - lo)
    return bounds, taps


def preprocess_frames_gpu(frames_u8: torch.Tensor, out_dtype=None, scale_size: int = 256, crop: int = 224) -> torch.Tensor:
    """frames uint8 [T,H,W,3] on the device -> [T,3,224,224] (the hot path's frame layout; the reference's
    ``load_video`` returns the same values as [3,T,224,224] fp32)."""
    out_dtype = runtime.HALF if out_dtype is None else out_dtype
    _chk(frames_u8, torch.uint8, "frames")
    T, H, Wd, C = frames_u8.shape
    assert C == 3
    nh, nw = resize_sizes(H, Wd, scale_size)
    if nh < crop or nw < crop:
        raise ValueError(f"Initial image size should be larger then cropped size but got cropped sizes : ({crop}, {crop}) "
                         f"while initial image is ({nw}, {nh})")
    x1, y1 = int(round((nw - crop) / 2.0)), int(round((nh - crop) / 2.0))
    d = frames_u8.device
    L = _lib.load()
    src = frames_u8
    if nw != Wd:                                             # horizontal pass, only the crop's columns
        bw, kw = resample_tables(Wd, nw)
        bw, kw = bw[x1:x1 + crop].copy(), kw[x1:x1 + crop].copy()
        tmp = torch.empty((T, H, crop, 3), dtype=torch.uint8, device=d)
        bw_d, kw_d = torch.from_numpy(bw).to(d), torch.from_numpy(kw).to(d)      # keep alive across the launch
        rc = L.vly_resize_h_u8(src.data_ptr(), bw_d.data_ptr(), kw_d.data_ptr(), tmp.data_ptr(), T, H, Wd, crop, kw.shape[1],
                               _stream())
        _lib.check(rc, "vly_resize_h_u8")
        src, x_off = tmp, 0
    else:
        x_off = x1
    if nh != H:
        bh, kh = resample_tables(H, nh)
        bh, kh = bh[y1:y1 + crop].copy(), kh[y1:y1 + crop].copy()
    else:                                                    # identity taps on the crop's rows
        bh = np.stack([np.arange(y1, y1 + crop), np.ones(crop)], 1).astype(np.int32)
        kh = np.full((crop, 1), 1 << PRECISION_BITS, np.int32)
    out = torch.empty((T, 3, crop, crop), dtype=out_dtype, device=d)
    bh_d, kh_d = torch.from_numpy(bh).to(d), torch.from_numpy(kh).to(d)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32, device=d)
    std = torch.tensor(CLIP_STD, dtype=torch.float32, device=d)
    rc = L.vly_resize_v_norm(src.data_ptr(), bh_d.data_ptr(), kh_d.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(),
                             T, src.shape[1], src.shape[2], x_off, crop, kh.shape[1], 1 if out_dtype == torch.float32 else 0, _stream())
    _lib.check(rc, "vly_resize_v_norm")
    return out
