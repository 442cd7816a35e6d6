#!/usr/bin/env python3
"""Where should the residual add of a sub-layer live?  Times, per shape, the two forms of "h += a @ W^T (+ b); x = norm(h)":
  A (the engines' form): GEMM -> 16-bit delta; add+norm kernel (h fp32 read + written, delta read, x written: 12 B per element)
  B: GEMM with the fp32 residual in its epilogue (h read + written by the GEMM), plain norm kernel (h read, x written: 6 B)
Both forms move the same bytes in total; B wins only if the GEMM hides its 8 B per element better than the norm kernel (which runs
at the HBM rate) does.  Weights rotate through NW copies so that every launch reads cold operands, as in the step.
Usage: resid_epilogue_ab.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops, runtime  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
d = torch.device("cuda:0")
HALF = runtime.HALF
SHAPES = [  # name, M, N, K, tile, layer norm?, bias?
    ("vit out-proj", 32768, 1024, 1024, 197, True, True),
    ("vit fc2", 32768, 1024, 4096, 197, True, True),
    ("13B o-proj", 2688, 5120, 5120, 198, False, False),
    ("13B down", 2688, 5120, 13824, 198, False, False),
]
only = os.environ.get("AB_ONLY")


def timed(fn, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    fn(0)
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        fn(i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return ts[len(ts) // 2]


for name, M, N, K, tile, ln, has_bias in SHAPES:
    if only and only not in name:
        continue
    NW = max(2, int(600e6 // (N * K * 2)))
    a = torch.randn((M, K), device=d).to(HALF)
    ws = [(torch.randn((N, K), device=d) * 0.03).to(HALF) for _ in range(NW)]
    packed = [ops.PackedWeight(w) for w in ws] if not ln else ws
    bias = torch.randn(N, device=d) if has_bias else None
    g = torch.rand(N, device=d) + 0.5
    b = torch.randn(N, device=d) * 0.1
    h0 = torch.randn((M, N), device=d)
    hA, hB = h0.clone(), h0.clone()
    delta = torch.empty((M, N), device=d, dtype=HALF)
    xA = torch.empty((M, N), device=d, dtype=HALF)
    xB = torch.empty((M, N), device=d, dtype=HALF)

    def gemm_a(i):
        ops.gemm_mfma(a, packed[i % NW], bias, out=delta, tile_hint=tile)

    def norm_a(i):
        ops.add_norm(hA, delta, g, b if ln else None, 1e-5, out=xA, rms=not ln)

    def gemm_b(i):
        ops.gemm_mfma(a, packed[i % NW], bias, residual=hB, out_dtype=torch.float32, out=hB, tile_hint=tile)

    def norm_b(i):
        if ln:
            ops.layernorm(hB, g, b, 1e-5, out=xB)
        else:
            ops.rmsnorm(hB, g, 1e-5, out=xB)

    # one pass of each form from the same state: the results must agree to the rounding of the 16-bit delta
    gemm_a(0); norm_a(0); gemm_b(0); norm_b(0)
    torch.cuda.synchronize()
    err = (hA - hB).abs().max().item()
    errx = (xA.float() - xB.float()).abs().max().item()
    tga, tna, tgb, tnb = timed(gemm_a, reps), timed(norm_a, reps), timed(gemm_b, reps), timed(norm_b, reps)

    def pair(f, gfn):
        return timed(lambda i: (f(i), gfn(i)), reps)

    tA, tB = pair(gemm_a, norm_a), pair(gemm_b, norm_b)
    print(f"{name:14s} {M}x{N}x{K} tile {tile}: A gemm {tga:7.1f} + add_norm {tna:6.1f} = pair {tA:7.1f} us | B gemm+resid {tgb:7.1f} + norm {tnb:6.1f}"
          f" = pair {tB:7.1f} us | B-A {tB - tA:+6.1f} us  (max |dh| {err:.2e}, |dx| {errx:.2e})", flush=True)
