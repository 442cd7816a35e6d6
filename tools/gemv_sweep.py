#!/usr/bin/env python3
"""GEMV (decode projection) rates on cold weights: rotates over enough weight copies to exceed the 256 MB
Infinity Cache.  gemv_sweep.py "N,K,epi,f32res;..." [M]  -> one JSON line per shape (TB/s of weight bytes); M rows (default 1)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

d = torch.device("cuda:0")
MROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for spec in sys.argv[1].split(";"):
    N, K, epi, f32res = (int(x) for x in spec.split(","))
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn((N, K), device=d) * 0.02).to(torch.bfloat16) for _ in range(ncopy)]
    a = torch.randn((MROWS, K), device=d).to(torch.bfloat16)
    No = N // 2 if epi == 2 else N
    res = torch.zeros((MROWS, N), device=d) if f32res else None
    out = res if f32res else torch.empty((MROWS, No), device=d, dtype=torch.bfloat16)
    for w in ws:
        ops.gemv(a, w, residual=res, epilogue=epi, out=out, out_dtype=out.dtype)
    torch.cuda.synchronize()
    reps = 5
    # one hipGraph of reps x copies launches: short kernels (N = 5120, K = 5120 runs ~8 us) would otherwise time the
    # ~10 us Python launch path, not the kernel
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(reps):
            for w in ws:
                ops.gemv(a, w, residual=res, epilogue=epi, out=out, out_dtype=out.dtype)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * ncopy)
    print(json.dumps({"M": MROWS, "N": N, "K": K, "epi": epi, "f32res": f32res, "us": round(us, 2),
                      "TBps": round(N * K * 2 / us / 1e6, 2), "copies": ncopy}), flush=True)
