"""Stream-K GEMM (vly_gemm_bf16_streamk) vs the whole-tile kernel and vs fp32 torch: same math, so
the two kernels may differ only by fp32 summation order (<= 1e-3 * sqrt(K) on O(1) data), never by
stale slab data (which would be O(1) errors).  Shapes cover: ranges cutting tiles, total < G,
K = 64 (one iteration per tile), every epilogue, repeated launches (epoch reuse of the workspace)."""
import math

import pytest
import torch

from valley_amd.runtime import HALF  # the library's 16-bit storage type: bf16, or fp16 under VALLEY_PRECISION=fp16 (this process is bound by the environment)

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


SHAPES = [(1312, 4096, 4096), (1312, 12288, 4096), (8224, 1024, 1024), (8224, 3072, 1024), (300, 264, 128),
          (64, 512, 64), (2688, 5120, 1024), (513, 4096, 640), (1028, 1024, 4096)]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 41, 42, 45, 51, 55, 73, 74, 76, 83, 84, 86, 102, 151, 176, 183, 184, 186])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_streamk_matches_tile_kernel(M, N, K, tile):
    from valley_amd import ops
    d = torch.device("cuda:0")
    a = rnd((M, K), 1, dtype=HALF).to(d)
    w = rnd((N, K), 2, 0.05, dtype=HALF).to(d)
    ref = ops.gemm_mfma(a, w, out_dtype=torch.float32)
    for rep in range(3):                                   # workspace reuse across epochs
        out = ops.gemm_streamk(a, w, out_dtype=torch.float32, tile_hint=tile)
        err = float((out - ref).abs().max())
        assert err < 1e-4 * math.sqrt(K), (rep, err)
    assert ops.sk_error_flag(d) == 0


@pytest.mark.parametrize("tile", [0, 1, 2, 5, 6, 51, 55, 73, 76, 83, 84, 86, 155, 186])
def test_streamk_epilogues(tile):
    from valley_amd import ops
    d = torch.device("cuda:0")
    M, N, K = 1312, 2048, 1024
    a = rnd((M, K), 3, dtype=HALF).to(d)
    w = rnd((N, K), 4, 0.05, dtype=HALF).to(d)
    bias = rnd((N,), 5, 0.5).to(d)
    res = rnd((M, N), 6).to(d)
    h = res.clone()
    ops.gemm_streamk(a, w, bias, residual=h, out=h, tile_hint=tile)            # in-place residual update
    ref = ops.gemm_mfma(a, w, bias, residual=res, out_dtype=torch.float32)
    assert float((h - ref).abs().max()) < 2e-3
    for epi in (ops.EPI_QUICK_GELU, ops.EPI_SWIGLU):
        b = bias if epi == ops.EPI_QUICK_GELU else None
        got = ops.gemm_streamk(a, w, b, epilogue=epi, tile_hint=tile).float()
        want = ops.gemm_mfma(a, w, b, epilogue=epi).float()
        assert float((got - want).abs().max()) <= 0.07      # one bf16 ulp at |x| < 8 where rounding flips
        assert float((got - want).norm() / want.norm()) < 2e-3
    assert ops.sk_error_flag(d) == 0


def test_streamk_stress_many_launches():
    """Back-to-back launches of different shapes sharing the workspace: stale flags/slabs must never leak."""
    from valley_amd import ops
    d = torch.device("cuda:0")
    mats = {}
    for (M, N, K) in SHAPES[:5]:
        a = rnd((M, K), M, dtype=HALF).to(d)
        w = rnd((N, K), N, 0.05, dtype=HALF).to(d)
        mats[(M, N, K)] = (a, w, ops.gemm_mfma(a, w, out_dtype=torch.float32))
    for rep in range(20):
        for (M, N, K), (a, w, ref) in mats.items():
            out = ops.gemm_streamk(a, w, out_dtype=torch.float32)
            assert float((out - ref).abs().max()) < 1e-4 * math.sqrt(K), (rep, M, N, K)
    assert ops.sk_error_flag(d) == 0
