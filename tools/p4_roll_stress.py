#!/usr/bin/env python3
"""Stress of the rolled persistent GEMM (round 5): the boundary's counted wait leans on vmcnt retiring LDS-DMA loads and stores in
issue order, and on the accumulators being nobody's but the asm's.  N launches per shape on rotating operands, EVERY result compared
bit for bit with the 16-wave tile kernel's (tile 9: same MFMA, same K order).  p4_roll_stress.py [launches]"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = torch.device("cuda:0")
torch.manual_seed(5)
bad = 0
for (M, N, K, epi, tile) in ((32896, 3072, 1024, 0, 198), (32768, 1024, 1024, 0, 197), (8200, 4360, 128, 0, 197), (32768, 4096, 1024, 1, 197),
                             (2688, 5120, 13824, 0, 198), (6000, 6104, 192, 3, 199)):
    As = [torch.randn((M, K), device=d).to(torch.bfloat16) for _ in range(3)]
    Ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(3)]
    bias = torch.randn((N,), device=d)
    refs = {}
    for i in range(n):
        a, w = As[i % 3], Ws[(i // 3) % 3]
        key = (i % 3, (i // 3) % 3)
        if key not in refs:
            refs[key] = ops.gemm_mfma(a, w, bias, epilogue=epi, tile_hint=9)
        got = ops.gemm_mfma(a, w, bias, epilogue=epi, tile_hint=tile)
        if not torch.equal(got, refs[key]):
            bad += 1
            print("MISMATCH", (M, N, K, epi, tile), "launch", i, float((got.float() - refs[key].float()).abs().max()), flush=True)
    torch.cuda.synchronize()
    print(f"{M}x{N}x{K}/e{epi} tile {tile}: {n} launches, mismatches so far {bad}", flush=True)
print("STRESS_OK" if bad == 0 else "STRESS_FAILED")
