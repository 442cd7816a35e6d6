"""Host-side video glue of the entry points (valley/util/data_util.py:40-56, 249-303).

``load_video`` needs ``decord`` to decode containers; this image has none, so the function accepts a
directory of frames / an ``.npy`` of uint8 frames [N,H,W,3] and otherwise raises.  The sampling and
normalisation follow the reference: 8 uniformly spaced frames (``np.linspace(0, len-1, 8).astype(int)``,
:264-265), short side to 256 (bilinear), centre crop 224, /255, CLIP mean/std (:272-273).
The GPU version of this preprocessing is the "next" row N2 of SURVEY.md §8(f)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import runtime

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class KeywordsStoppingCriteria:
    """valley/util/data_util.py:40-56."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.tokenizer = tokenizer
        self.start_len = None
        self.input_ids = input_ids

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        if self.start_len is None:
            self.start_len = self.input_ids.shape[1]
        else:
            outputs = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
            for keyword in self.keywords:
                if keyword in outputs:
                    return True
        return False


def sample_indices(n_total: int, n: int = 8) -> np.ndarray:
    return np.linspace(0, n_total - 1, n).astype(np.int_)


def preprocess_frames(frames_u8: np.ndarray, scale_size: int = 256, crop: int = 224) -> torch.Tensor:
    """uint8 [T,H,W,3] -> float32 [3,T,224,224], the tensor ``load_video`` returns (valley/util/data_util.py:262-281):
    short side to 256 with PIL's BILINEAR resample (which antialiases when it shrinks; target size floor-rounded,
    valley/data/video_transform.py:74-81), centre crop 224 (offset ``int(round((w - 224) / 2.))``, :542-544), /255,
    CLIP mean/std.  With a GPU the frames go through the Pillow-exact HIP kernels (valley_amd/preprocess.py) and the
    result stays on the device; without one this is the reference's own host chain, PIL itself doing the resample —
    both entry points of the model therefore see the same pixels (tests/test_preprocess_cpu.py)."""
    frames_u8 = np.ascontiguousarray(frames_u8)
    if torch.cuda.is_available():
        from .preprocess import preprocess_frames_gpu
        out = preprocess_frames_gpu(torch.from_numpy(frames_u8).cuda(), out_dtype=torch.float32, scale_size=scale_size, crop=crop)
        return out.permute(1, 0, 2, 3).contiguous()
    from PIL import Image

    from .preprocess import resize_sizes
    T, H, W, _ = frames_u8.shape
    nh, nw = resize_sizes(H, W, scale_size)
    if nh < crop or nw < crop:
        raise ValueError(f"Initial image size should be larger then cropped size but got cropped sizes : ({crop}, {crop}) "
                         f"while initial image is ({nw}, {nh})")
    x1, y1 = int(round((nw - crop) / 2.0)), int(round((nh - crop) / 2.0))
    out = np.empty((T, crop, crop, 3), np.float32)
    for t in range(T):
        im = Image.fromarray(frames_u8[t])
        if (nh, nw) != (H, W):
            im = im.resize((nw, nh), Image.BILINEAR)
        out[t] = np.asarray(im, dtype=np.float32)[y1:y1 + crop, x1:x1 + crop]
    x = torch.from_numpy(out).permute(3, 0, 1, 2) / 255.0                                  # [3,T,224,224]
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1, 1)
    return ((x - mean) / std).contiguous()


def load_video(path, fixed_frame_number: int = 8) -> torch.Tensor:
    if isinstance(path, np.ndarray):
        frames = path
    elif str(path).endswith(".npy"):
        frames = np.load(path)
    elif os.path.isfile(path):
        try:
            import decord
        except ImportError as e:
            raise RuntimeError("decoding video containers needs `decord`, which this image does not ship; pass a "
                               "preprocessed [3,T,224,224] tensor, a uint8 frame array or an .npy file") from e
        vr = decord.VideoReader(path, num_threads=1, ctx=decord.cpu(0))
        frames = vr.get_batch(sample_indices(len(vr), fixed_frame_number)).asnumpy()
        return preprocess_frames(frames)
    else:
        raise FileNotFoundError(path)
    return preprocess_frames(frames[sample_indices(len(frames), fixed_frame_number)])


def load_video_gpu(frames_u8, device="cuda:0", fixed_frame_number: int = 8, dtype=None) -> torch.Tensor:
    """Decoded frames uint8 [N,H,W,3] (numpy or tensor) -> the hot path's input [1,T,3,224,224] on the
    device: uniform sampling of ``fixed_frame_number`` frames (data_util.py:264-265), then the GPU
    preprocessing kernels (valley_amd/preprocess.py; Pillow-exact resize, crop, normalise)."""
    from .preprocess import preprocess_frames_gpu
    if isinstance(frames_u8, np.ndarray):
        frames_u8 = torch.from_numpy(np.ascontiguousarray(frames_u8))
    idx = torch.from_numpy(sample_indices(frames_u8.shape[0], fixed_frame_number))
    sel = frames_u8[idx].to(device).contiguous()
    return preprocess_frames_gpu(sel, out_dtype=dtype).unsqueeze(0)
