#!/usr/bin/env python3
"""Time the skinny GEMM at the row-split remainders (128 rows)."""
import os, statistics, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops
d = torch.device("cuda:0")
for (M, N, K, epi) in [(128, 1024, 4096, 0), (128, 1024, 1024, 0), (128, 4096, 1024, 1), (256, 1024, 4096, 0)]:
    a = torch.randn((M, K), device=d).to(torch.bfloat16); ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.randn(N, device=d)
    ref = (a.float() @ ws[0].float().t() + bias)
    if epi == 1: ref = ref * torch.sigmoid(1.702 * ref)
    got = ops.gemm_skinny(a, ws[0], bias, epilogue=epi)
    err = float((got.float() - ref).norm() / ref.norm())
    ts = []
    for r in range(43):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_skinny(a, ws[r % 4], bias, epilogue=epi); e1.record(); torch.cuda.synchronize()
        if r >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"shape": f"{M}x{N}x{K}/e{epi}", "us": round(statistics.median(ts), 1), "relerr": round(err, 5)}), flush=True)
