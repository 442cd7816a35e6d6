#!/usr/bin/env python3
"""Per-op timing of the temporal pooling variants at production width (H = 4096, T = 8, B = 2..8) — N4 of SURVEY §8f."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valley_amd import ops

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

d = "cuda:0"
for B in (2, 8):
    T, H = 8, 4096
    F = B * T
    feats = torch.randn((F * 257, 1024), device=d)
    w = (torch.randn((H, 1024), device=d) * 0.03).to(torch.bfloat16)
    bias = torch.zeros(H, device=d)
    pw, pb = torch.randn(256 * H, device=d) * 0.002, torch.zeros(1, device=d)
    x16 = ops.cast_bf16(feats)
    proj = ops.gemm_mfma(x16, w, bias, out_dtype=torch.float32)
    sc = ops.temporal_scores(proj, pw, pb, F)
    print(f"B={B}: cast {t(lambda: ops.cast_bf16(feats)):.1f}us  project-all {t(lambda: ops.gemm_mfma(x16, w, bias, out_dtype=torch.float32)):.1f}us  "
          f"scores {t(lambda: ops.temporal_scores(proj, pw, pb, F)):.1f}us  pool-importance {t(lambda: ops.pool_tokens(proj, B, T, ops.POOL_IMPORTANCE, sc)):.1f}us  "
          f"pool-max {t(lambda: ops.pool_tokens(proj, B, T, ops.POOL_MAX)):.1f}us  pool-mean(1024) {t(lambda: ops.pool_tokens(feats, B, T, ops.POOL_MEAN)):.1f}us")
