#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc runs): gemm_one.py M N K tile|sk|lib|skinny tile [epi] [reps] [packed]
(packed = 1: the weight in the block-ordered copy the prefill engines read, ops.PackedWeight)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind, tile = sys.argv[4], int(sys.argv[5])
epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
packed = len(sys.argv) > 8 and sys.argv[8] == "1"
d = torch.device("cuda:0")
a = torch.randn((M, K), device=d).to(torch.bfloat16)
w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
if packed:
    w = ops.PackedWeight(w)
fn = ops.gemm_mfma if kind == "tile" else ops.gemm_streamk
for _ in range(reps):
    if kind == "skinny":
        ops.gemm_skinny(a, w, None, epi, out)
    elif kind == "lib":          # the vendor library's kernel for the same problem (information only: its counters beside ours)
        torch.mm(a, w.t(), out=out)
    else:
        fn(a, w, epilogue=epi, out=out, tile_hint=tile)
torch.cuda.synchronize()
