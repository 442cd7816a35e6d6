"""Worker of tests/test_parallel_gpu.py: one of WORLD_SIZE ranks sharing cuda:0 (a 1-GPU box), backend gloo — or, with
VALLEY_DP_BACKEND=nccl on a box with >= WORLD_SIZE GPUs, one GPU per rank with RCCL carrying the all-gather.  Runs the
REAL encoder (HipCLIPVisionTower + pooling) through valley_amd.parallel and compares, bit for bit, with the single-rank
result it computes itself (SURVEY.md §8e "Verification")."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("VALLEY_DP_BACKEND", "gloo")
    if backend == "nccl":                                    # one GPU per rank, RCCL over xGMI carries the gather
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    else:
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    assert dist.get_backend() == backend
    from tests import golden_cfg as G
    from tests.test_model_gpu import build_golden_model
    from valley_amd import ops, parallel
    ops.GEMM_MODE = "tiles"                                  # batch-invariant kernels: shard results == batch results
    for method in ("mean", "max"):
        model = build_golden_model(method)
        mm = model.get_model()
        px = torch.from_numpy(G.golden_pixels(12, "dp")).cuda()    # (the current device: cuda:0, or this rank's own GPU)
        # 1. equal clips: [4 clips x 3 frames]
        clips = [px[i * 3:(i + 1) * 3] for i in range(4)]
        want, Ts = mm.encode_clips(clips)
        got, Ts2 = parallel.encode_clips_dp(mm.encode_clips, clips)
        assert Ts == Ts2 and torch.equal(got, want), (method, "equal clips")
        # 2. ragged T: clips of 2, 5, 1, 4 frames (shards of different row counts -> padded gather)
        cuts = [0, 2, 7, 8, 12]
        clips = [px[cuts[i]:cuts[i + 1]] for i in range(4)]
        want, _ = mm.encode_clips(clips)
        got, _ = parallel.encode_clips_dp(mm.encode_clips, clips)
        assert torch.equal(got, want), (method, "ragged")
        # 3. fewer clips than ranks: the idle rank contributes an empty [0, W] shard (W = 1024 for mean, H for max)
        clips = [px[:4]]
        want, _ = mm.encode_clips(clips)
        got, _ = parallel.encode_clips_dp(mm.encode_clips, clips)
        assert torch.equal(got, want) and got.shape[1] == (1024 if method == "mean" else G.GCFG["H"]), (method, "idle rank")
    # 4. frames mode: frames of one clip split across ranks, fp32 features gathered
    tower = mm.vision_tower
    want = tower.encode(px[:5], -2)
    got = parallel.encode_frames_dp(lambda f: tower.encode(f, -2), px[:5])
    assert torch.equal(got, want), "frames mode"
    dist.barrier()
    print(f"DP_OK rank {rank}/{world} backend {dist.get_backend()} device {torch.cuda.current_device()}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
