"""CPU oracle for the frame preprocessing that feeds the hot path.  TEST
INFRASTRUCTURE ONLY (same rule as valley_oracle.py).

Restates what ``load_video`` does to decoded frames (valley/util/data_util.py:262-281):
    TensorToNumpy -> PIL images                     valley/data/video_transform.py:744-752
    Resize(256): short side to 256, PIL BILINEAR    :33-82, 259-274  (the 'nearest' default selects
                                                    PIL.Image.BILINEAR because of the inverted test at :63-66)
    CenterCrop(224)                                 :505-546 (x1 = int(round((w-224)/2.)))
    ClipToTensor: /255, [3,T,H,W] float32           :113-164
    Normalize(CLIP mean/std)                        :84-100, 715-741
The resize arithmetic lives in the third-party Pillow (not vendored; any Pillow >= 3.x uses this
algorithm): libImaging/Resample.c — separable convolution, horizontal pass then vertical pass, 8-bit
fixed point: coefficients (int)(k * 2^22 +- 0.5), accumulate from 2^21, clip8(ss >> 22), the
intermediate image is uint8.  ``resample_coeffs`` follows precompute_coeffs()/normalize_coeffs_8bpc().
Pinned in tests/test_preprocess_cpu.py against PIL itself (bit-exact) and against a fixture captured from
the reference's own transform classes (tools/gen_preprocess_goldens.py)."""
import math

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2


def get_resize_sizes(im_h, im_w, size):
    """video_transform.py:74-81."""
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)


def resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs (bilinear: support 1.0) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        n = xmax - xmin
        x = np.arange(n)
        w = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :n] = w
        bounds[xx] = (xmin, n)
    ki = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return bounds, ki.astype(np.int32)


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """one 8-bit resampling pass along ``axis`` of a uint8 [H,W,C] image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[i, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_bilinear_resize(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [new_h,new_w,3], bit-exact with PIL.Image.resize(..., BILINEAR)."""
    H, W, _ = img.shape
    out = img
    if new_w != W:
        b, k = resample_coeffs(W, new_w)
        out = _pass(out, b, k, axis=1)
    if new_h != H:
        b, k = resample_coeffs(H, new_h)
        out = _pass(out, b, k, axis=0)
    return out


def preprocess_frames(frames_u8: np.ndarray, scale_size: int = 256, crop: int = 224) -> np.ndarray:
    """uint8 [T,H,W,3] -> float32 [3,T,224,224]: the tensor ``load_video`` returns."""
    T, H, W, _ = frames_u8.shape
    if (W <= H and W == scale_size) or (H <= W and H == scale_size):
        nh, nw = H, W
    else:
        nh, nw = get_resize_sizes(H, W, scale_size)
    x1 = int(round((nw - crop) / 2.0))
    y1 = int(round((nh - crop) / 2.0))
    out = np.empty((3, T, crop, crop), np.float32)
    mean = np.asarray(CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(CLIP_STD, np.float32)[:, None, None]
    for t in range(T):
        r = pil_bilinear_resize(frames_u8[t], nh, nw) if (nh, nw) != (H, W) else frames_u8[t]
        c = r[y1:y1 + crop, x1:x1 + crop].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        out[:, t] = (c - mean) / std
    return out
