#!/usr/bin/env python3
"""Interleaved timing of GEMM kernel choices on one shape: gemm_ab.py "M,N,K,epi:197,s297,...;..." — a bare number is a
vly_gemm_bf16 tile hint, s<number> a vly_gemm_bf16_streamk hint.  Activations rotate through 3 copies and weights (packed
copies, as the engines hold them) through enough copies to leave the caches; one JSON line per shape (median us, TFLOP/s)."""
import json
import os
import random
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

d = torch.device("cuda:0")
rng = random.Random(0)
for spec in sys.argv[1].split(";"):
    shape, cands = spec.split(":")
    M, N, K, epi = (int(x) for x in shape.split(","))
    cands = cands.split(",")
    As = [torch.randn((M, K), device=d).to(torch.bfloat16) for _ in range(3)]
    ncopy = max(2, int(300e6 // (N * K * 2)) + 1)
    Ws = [ops.PackedWeight((torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)) for _ in range(ncopy)]
    out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)

    def run(c, i):
        if c.startswith("s"):
            ops.gemm_streamk(As[i % 3], Ws[i % ncopy], None, epilogue=epi, out=out, tile_hint=int(c[1:]))
        else:
            ops.gemm_mfma(As[i % 3], Ws[i % ncopy], None, epilogue=epi, out=out, tile_hint=int(c))
    times = {c: [] for c in cands}
    n = 0
    for r in range(23):
        order = list(cands)
        rng.shuffle(order)
        for c in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(c, n)
            e1.record()
            n += 1
            torch.cuda.synchronize()
            if r >= 3:
                times[c].append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"shape": f"{M}x{N}x{K}/e{epi}", **{c: {"us": round(statistics.median(t), 1), "TFLOPs": round(2.0 * M * N * K / statistics.median(t) / 1e6, 1)}
                                                          for c, t in times.items()}, "sk_error_flag": ops.sk_error_flag(d)}), flush=True)
