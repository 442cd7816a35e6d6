"""world_size-2 gloo tests of the frame-DP sharding + all-gather (valley_amd/parallel.py): the gathered
tensor must be bit-identical to the single-rank concat of the same clips."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from valley_amd import parallel as P


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            spans = [P.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == P.shard_sizes(n, w)


def fake_encode_pool(clips):
    """Deterministic stand-in for tower+pool on CPU: rows depend only on the clip's content."""
    outs, Ts = [], []
    for c in clips:
        T = c.shape[0]
        base = c.float().mean(dim=(1, 2, 3))                          # [T]
        pooled = base.mean() + torch.arange(256, dtype=torch.float32)[:, None] * 0.01 + torch.zeros(256, 1024)
        cls = base[:, None] + torch.zeros(T, 1024)
        outs.append(torch.cat([pooled, cls], 0).to(torch.bfloat16))
        Ts.append(T)
    return torch.cat(outs, 0), Ts


def _worker(rank, world, port, Ts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        clips = [torch.randn((t, 3, 8, 8), generator=g) for t in Ts]
        got, ts = P.encode_clips_dp(fake_encode_pool, clips)
        ref, _ = fake_encode_pool(clips)
        frames = torch.cat(clips, 0)
        feats = P.encode_frames_dp(lambda f: f.flatten(1)[:, :16].clone(), frames)
        q.put((rank, bool(torch.equal(got, ref)), ts == Ts, bool(torch.equal(feats, frames.flatten(1)[:, :16]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Ts", [[4, 4, 4, 4], [2, 3, 5], [8]])
def test_two_rank_gather_equals_single_rank_concat(Ts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + sum(Ts) * 7 + len(Ts)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, Ts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and tsok and fok for _, ok, tsok, fok in res), res
