"""Round 6 GPU parity: the persistent GEMMs with accumulators by name on the 32x32x16 MFMA (tile hints 397 / 398, gemm_p32.hip) and on
16x16x32 in chains of two (497, gemm_p16.hip) against fp32 PyTorch and against the default persistent kernel (197) — every epilogue
they take, bias through the matrix cores / from the LDS strip, block-packed weights, ragged M and N edges, more tiles than CUs, a wave
slab outside the problem, asymmetric data (a transposed fragment cannot hide), guard columns; what they do not take falls back to 197.
And the C ABI of the split-operand images (vly_split3_f32, vly_norm_split3_f32)."""
import math

import pytest
import torch

from valley_amd.runtime import HALF

pytestmark = pytest.mark.gpu
D = "cuda:0"
HINTS = [397, 398, 497]


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def ref_of(a, w, bias, epi):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    elif epi == 3:
        y = torch.relu(y)
    elif epi == 2:
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    return y


def relerr(got, ref):
    return float((got.float() - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("M,N,K,epi,has_bias,packed", [
    (256, 256, 576, 0, False, False),          # one tile, the shortest K the kernels take (9 K tiles)
    (512, 512, 640, 0, True, False),           # bias, one tile per workgroup, tiles_n = 2 (n-fastest order: group height 1)
    (700, 1024, 1024, 0, True, False),         # last m-tile: a wave slab outside the problem
    (1000, 1032, 704, 0, True, False),         # ragged N (1032 = 4 x 256 + 8) and M
    (2056, 4096, 1024, 1, True, False),        # quick_gelu, ViT fc1 at 8 frames
    (2056, 1024, 4096, 0, True, True),         # block-packed weights, ViT fc2
    (3000, 3072, 1024, 3, True, False),        # ReLU
    (2688, 2048, 1024, 2, False, True),        # SwiGLU, packed
    (336, 32008, 1024, 0, False, False),       # lm_head: 252 tiles, ragged N
    (9000, 3072, 1024, 0, True, False),        # 432 tiles on 256 workgroups: tile boundaries, parked stores under the next tile
])
def test_named_accumulator_gemms_vs_fp32_and_default(M, N, K, epi, has_bias, packed, hint):
    from valley_amd import ops
    a = rnd((M, K), 1, dtype=HALF).to(D)
    w = rnd((N, K), 2, 0.05, dtype=HALF).to(D)
    bias = rnd((N,), 3).to(D) if has_bias else None
    wp = ops.PackedWeight(w) if packed else w
    ref = ref_of(a, w, bias, epi)
    base = ops.gemm_mfma(a, wp, bias, epilogue=epi, tile_hint=197)
    out = ops.gemm_mfma(a, wp, bias, epilogue=epi, tile_hint=hint)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    e, e0 = relerr(out, ref), relerr(base, ref)
    assert e < 4e-3 and e <= 1.05 * e0 + 1e-5, (e, e0)
    # same 16-bit rounding of nearly the same fp32 sums: the two kernels differ in a few last bits at most
    assert float((out.float() - base.float()).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())


@pytest.mark.parametrize("hint", HINTS)
def test_named_accumulator_gemms_asymmetric_and_guard_columns(hint):
    """One-hot weights (a swapped fragment or a transposed block lands in the wrong cell) on a strided A; output into a wider buffer:
    nothing may be written past column N, nothing past row M."""
    from valley_amd import ops
    M, N, K = 777, 520, 640
    big = rnd((M + 5, K + 128), 7, dtype=HALF).to(D)
    a = big[:M, 64:64 + K]
    w = torch.zeros((N, K), dtype=HALF)
    for n, k, v in ((3, 17, 1.0), (100, 639, 2.0), (519, 0, -1.0), (256, 320, 0.5), (31, 63, 4.0)):
        w[n, k] = v
    out = torch.full((M + 3, N + 24), 7.0, dtype=HALF, device=D)
    ops.gemm_mfma(a, w.to(D), out=out[:M, :N], tile_hint=hint)
    torch.cuda.synchronize()
    ref = (a.float().cpu() @ w.float().t()).to(HALF)
    assert torch.equal(out[:M, :N].cpu(), ref)
    assert float((out[:, N:].float() - 7.0).abs().max()) == 0.0 and float((out[M:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("hint", HINTS)
def test_named_accumulator_gemms_fall_back_where_they_do_not_apply(hint):
    """fp32 output, a residual, K below nine K tiles, an unaligned output row stride: the hint resolves to the default persistent kernel
    (or its tile kernel) and the results are bit-identical to hint 197's."""
    from valley_amd import ops
    a = rnd((600, 512), 11, dtype=HALF).to(D)
    w = rnd((768, 512), 12, 0.05, dtype=HALF).to(D)
    res = rnd((600, 768), 13).to(D)
    for kw in (dict(out_dtype=torch.float32), dict(residual=res, out_dtype=torch.float32)):
        assert torch.equal(ops.gemm_mfma(a, w, tile_hint=hint, **kw), ops.gemm_mfma(a, w, tile_hint=197, **kw))
    o1 = torch.empty((600, 770), dtype=HALF, device=D)
    o2 = torch.empty((600, 770), dtype=HALF, device=D)
    ops.gemm_mfma(a, w, out=o1[:, :768], tile_hint=hint)               # ldc % 8 != 0
    ops.gemm_mfma(a, w, out=o2[:, :768], tile_hint=197)
    assert torch.equal(o1[:, :768], o2[:, :768])


def test_named_accumulator_gemms_are_deterministic_across_launches():
    from valley_amd import ops
    a = rnd((4000, 1024), 21, dtype=HALF).to(D)
    w = rnd((3072, 1024), 22, 0.05, dtype=HALF).to(D)
    bias = rnd((3072,), 23).to(D)
    for hint in HINTS:
        outs = [ops.gemm_mfma(a, w, bias, epilogue=1, tile_hint=hint).clone() for _ in range(4)]
        assert all(torch.equal(outs[0], o) for o in outs[1:]), hint


@pytest.mark.parametrize("M,K,epi,order", [(300, 128, 0, 0), (257, 592, 0, 0), (64, 1024, 1, 0), (100, 512, 3, 0), (96, 256, 2, 0), (520, 320, 0, 1)])
def test_split3_images(M, K, epi, order):
    """vly_split3_f32 through the C ABI: [hi | hi | lo] (activations) / [hi | lo | hi] (weights), hi = rn16(x), lo = rn16(x - hi), pad columns
    zero, the producing GEMM's activation applied first — against the same arithmetic in PyTorch, bit for bit on hi and lo."""
    from valley_amd import lib as _lib
    L = _lib.load()
    kin = 2 * K if epi == 2 else K
    x = rnd((M, kin), 31, 2.0).to(D)
    Kp = (K + 63) // 64 * 64
    out = torch.full((M, 3 * Kp), 9.0, dtype=HALF, device=D)
    rc = L.vly_split3_f32(x.data_ptr(), kin, out.data_ptr(), M, K, Kp, epi, order, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    if epi == 1:
        y = x / (1 + torch.exp(-1.702 * x))
    elif epi == 3:
        y = torch.relu(x)
    elif epi == 2:
        y = x[:, 0::2] / (1 + torch.exp(-x[:, 0::2])) * x[:, 1::2]
    else:
        y = x
    hi = out[:, :K].float()
    lo = out[:, 2 * Kp:2 * Kp + K].float() if order == 0 else out[:, Kp:Kp + K].float()
    hi2 = out[:, Kp:Kp + K].float() if order == 0 else out[:, 2 * Kp:2 * Kp + K].float()
    assert torch.equal(hi, hi2)
    # hi + lo reproduces the fp32 value to 2^-16 relative; with epi == 0 the split itself is checked bit for bit
    assert float(((hi + lo) - y).abs().max()) <= 2.0 ** -15 * float(y.abs().max()) + 1e-30
    if epi == 0:
        assert torch.equal(hi, y.to(HALF).float()) and torch.equal(lo, (y - y.to(HALF).float()).to(HALF).float())
    if Kp > K:
        for seg in range(3):
            assert float(out[:, seg * Kp + K:(seg + 1) * Kp].float().abs().max()) == 0.0


@pytest.mark.parametrize("D_,rms", [(1024, 0), (5120, 1), (320, 1)])
def test_norm_split3_equals_norm_then_split(D_, rms):
    from valley_amd import lib as _lib, ops_f32 as F
    L = _lib.load()
    M = 70
    x = rnd((M, D_), 41, 3.0).to(D)
    g, b = rnd((D_,), 42).to(D), rnd((D_,), 43).to(D)
    y = F.norm(x, g, None if rms else b, 1e-5)
    Kp = (D_ + 63) // 64 * 64
    a3 = torch.empty((M, 3 * Kp), dtype=HALF, device=D)
    st = torch.cuda.current_stream().cuda_stream
    assert L.vly_split3_f32(y.data_ptr(), D_, a3.data_ptr(), M, D_, Kp, 0, 0, st) == 0
    b3 = torch.empty_like(a3)
    assert L.vly_norm_split3_f32(x.data_ptr(), g.data_ptr(), None if rms else b.data_ptr(), b3.data_ptr(), M, D_, Kp, 1e-5, rms, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(a3, b3)
