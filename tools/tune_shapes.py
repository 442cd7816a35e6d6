#!/usr/bin/env python3
"""Run the online GEMM tuner on explicit shapes (those that only appear at N > 1: the projector GEMM sees
N x B x (256+T) rows) and add the decisions to a tuning table:  tune_shapes.py table.json "M,N,K,bias;..." """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VALLEY_TUNE_CACHE"] = sys.argv[1]
from valley_amd import ops  # noqa: E402

d = torch.device("cuda:0")
for spec in sys.argv[2].split(";"):
    M, N, K, hb = (int(x) for x in spec.split(","))
    a = torch.randn((M, K), device=d).to(torch.bfloat16)
    w = (torch.randn((N, K), device=d) * 0.02).to(torch.bfloat16)
    b = torch.randn((N,), device=d) if hb else None
    key = ops._tune_key(M, N, K, ops.EPI_NONE, torch.bfloat16, bool(hb), False, None)
    n = 0
    while key not in ops._TUNED and n < 1000:
        ops.gemm(a, w, b)
        torch.cuda.synchronize()
        n += 1
    print(spec, "->", ops._TUNED.get(key), "after", n, "calls", flush=True)
