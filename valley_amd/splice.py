"""Host-side integer logic of the visual-token splice (valley/model/valley_model.py:195-247).

The reference decides, per sample, which rows of ``inputs_embeds`` are replaced by pooled patch
tokens (after each <im_start>) and by per-frame CLS tokens (after each <vi_start>), raising
ValueError for unbalanced / cut image blocks and silently keeping the image-only splice when the
video block is malformed.  That is pure index arithmetic on ``input_ids``; it stays on the host
(B*S integers) and produces a row map for the device gather kernel ``vly_embed_splice``:

    row_map[b*S + s] =  token id            -> row of the embedding table
                     = -(visual_row + 1)    -> row of the projected visual-token buffer

Visual buffer layout: clip c occupies rows [off_c, off_c + P + T_c): P pooled patch rows, then T_c
CLS rows, off_c = sum_{c' < c} (P + T_c').
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def build_row_map(input_ids: np.ndarray, frames_per_clip: Sequence[int], tok, n_patches: int = 256) -> np.ndarray:
    """input_ids int [B,S]; frames_per_clip[c] = T of the c-th clip in ``images``; ``tok`` carries the
    six ids bound onto vision_tower.config (run_valley.py:13-18).  Returns int32 [B*S]."""
    ids = np.asarray(input_ids)
    B, S = ids.shape
    P = n_patches
    offs = np.concatenate([[0], np.cumsum([P + t for t in frames_per_clip])]).astype(np.int64)
    row_map = ids.astype(np.int32).copy()
    cur_image_idx = 0                                        # advances only on multimodal samples (:246)
    for b in range(B):
        cur = ids[b]
        if (cur == tok.im_patch_token).sum() == 0:           # :198-202 text-only sample
            continue
        if cur_image_idx >= len(frames_per_clip):
            raise IndexError("index %d is out of bounds for image_features of size %d"
                             % (cur_image_idx, len(frames_per_clip)))
        T = int(frames_per_clip[cur_image_idx])
        base = int(offs[cur_image_idx])
        if (cur == tok.im_start_token).sum() != (cur == tok.im_end_token).sum():      # :219-220
            raise ValueError("The number of im_start_token and im_end_token should be the same")
        img_map = row_map[b].copy()
        for pos in np.where(cur == tok.im_start_token)[0]:
            pos = int(pos)
            if pos + P + 1 >= S:                             # reference indexes unguarded
                raise IndexError(f"index {pos + P + 1} is out of bounds for dimension 0 with size {S}")
            if cur[pos + P + 1] != tok.im_end_token:         # :226-227
                raise ValueError("Seems that the image is cut.")
            img_map[pos + 1: pos + P + 1] = -(base + np.arange(P, dtype=np.int32) + 1)
        vid_map = img_map.copy()
        try:                                                 # :231-244, bare except
            if (cur == tok.vi_start_token).sum() != (cur == tok.vi_end_token).sum():
                raise ValueError("The number of vi_start_token and vi_end_token should be the same")
            assert (cur == tok.vi_frame_token).sum() == T
            for pos in np.where(cur == tok.vi_start_token)[0]:
                pos = int(pos)
                if cur[pos + T + 1] != tok.vi_end_token:
                    raise ValueError("Seems that the image is cut.")
                vid_map[pos + 1: pos + T + 1] = -(base + P + np.arange(T, dtype=np.int32) + 1)
        except Exception:                                    # noqa: BLE001 - mirrors the reference
            vid_map = img_map
        row_map[b] = vid_map
        cur_image_idx += 1
    return row_map.reshape(-1)
