"""Frame data parallelism across the GPUs of one node (BASELINE.json north_star).

The reference has no inference-time collective; its natural shard is the CLIP frame: nothing couples
two frames until temporal pooling, and pooling couples only the T frames of ONE clip.  So:

* ``clips`` mode (default, B >= world): rank r encodes whole clips [start_r, end_r), pools them
  locally, and ONE all-gather reassembles the pooled visual tokens — pre-projection, 1024 wide
  bf16, (256+T)*2 KiB per clip — on every rank before the LLM step.  The projector then runs on the
  gathered tokens (replicated, 2*(256+T)*1024*H flop per clip: trivial).
* ``frames`` mode (B < world): frames of each clip are split across ranks, the fp32 features
  [F_local, 257, 1024] are all-gathered and pooled on every rank.

One process per GPU; the collective is ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  Payloads are small (4.7 MB/rank at config 4), so the all-gather is
latency-, not bandwidth-bound, and a single call per step is the right granularity.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import runtime


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n items: the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def all_gather_rows(local: torch.Tensor, rows_per_rank: Sequence[int], group=None) -> torch.Tensor:
    """All-gather along dim 0 with (possibly) unequal row counts.  Equal shards take the single
    ``all_gather_into_tensor`` fast path; ragged ones are padded to the largest shard."""
    world = dist.get_world_size(group)
    assert len(rows_per_rank) == world and local.shape[0] == rows_per_rank[dist.get_rank(group)]
    tail = tuple(local.shape[1:])
    mx = max(rows_per_rank)
    if all(r == mx for r in rows_per_rank):
        out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    buf = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + rows_per_rank[r]] for r in range(world)], 0)


def encode_clips_dp(encode_pool: Callable[[list], Tuple[torch.Tensor, List[int]]], clips: Sequence[torch.Tensor],
                    group=None, width: Optional[int] = None, device=None) -> Tuple[torch.Tensor, List[int]]:
    """``clips`` mode.  ``encode_pool(list_of_clips) -> (pooled [sum(256+T_i), W], Ts)`` is the local
    encoder (``ValleyLlamaModel.encode_clips``).  Every rank passes the SAME full clip list (or at
    least agrees on the frame counts); returns pooled tokens of ALL clips in order, on every rank.
    W is 1024 for mean pooling and H for the variants that project before pooling; a rank without clips (more ranks
    than clips) contributes an empty [0, W] shard — W and the device come from ``width`` / ``device`` when given, else W
    is agreed with one tiny MAX all-reduce (taken by every rank, only when some rank is idle) and the device is the
    rank's current accelerator (CPU tensors for CPU clips under gloo)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    Ts = [int(c.shape[0]) for c in clips]
    s, e = shard_range(len(clips), rank, world)
    rows = []
    for r in range(world):
        a, b = shard_range(len(clips), r, world)
        rows.append(sum(256 + t for t in Ts[a:b]))
    local = encode_pool(list(clips[s:e]))[0] if e > s else None
    if min(rows) == 0:                                       # every rank sees the same `rows`: same branch everywhere
        if device is None:
            device = local.device if local is not None else \
                (torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and clips[0].is_cuda or
                                                                        dist.get_backend(group) == "nccl") else torch.device("cpu"))
        if width is None:
            wt = torch.tensor([0 if local is None else local.shape[1]], dtype=torch.int64, device=device)
            dist.all_reduce(wt, op=dist.ReduceOp.MAX, group=group)
            width = int(wt.item())
        if local is None:
            local = torch.empty((0, width), dtype=runtime.HALF, device=device)
    return all_gather_rows(local, rows, group), Ts


def encode_frames_dp(encode_frames: Callable[[torch.Tensor], torch.Tensor], frames: torch.Tensor, group=None) -> torch.Tensor:
    """``frames`` mode: frames [F,3,224,224] split across ranks, features [F,257,W] gathered everywhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    F = frames.shape[0]
    s, e = shard_range(F, rank, world)
    local = encode_frames(frames[s:e]) if e > s else None
    if local is None:
        probe = encode_frames(frames[:1])
        local = probe[:0]
    return all_gather_rows(local, shard_sizes(F, world), group)
