#!/usr/bin/env python3
"""q|k|v projection of the 13B prefill (8 x 336 rows): vly_gemm_bf16 + vly_rope_kv against the fused epilogue, per tile."""
import os, statistics, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops
d = torch.device("cuda:0")
B, S, heads, K, ctx = 8, 336, 40, 5120, 640
H = heads * 128
a = [torch.randn((B * S, K), device=d).to(torch.bfloat16) for _ in range(3)]
w = [(torch.randn((3 * H, K), device=d) * 0.02).to(torch.bfloat16) for _ in range(4)]
pos = torch.arange(ctx, device=d, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, 128, 2, device=d, dtype=torch.float32) / 128))[None, :]
cos, sin = pos.cos().contiguous(), pos.sin().contiguous()
kc = torch.zeros((B, heads, ctx, 128), device=d, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
qkv = torch.empty((B * S, 3 * H), device=d, dtype=torch.bfloat16)
def timed(fn):
    ts = []
    for r in range(23):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(r); e1.record(); torch.cuda.synchronize()
        if r >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 1)
res = {}
for t in (198, 98, 197):
    def unf(r):
        ops.gemm_mfma(a[r % 3], w[r % 4], out=qkv, tile_hint=t)
        ops.rope_kv(qkv, kc, vc, cos, sin, B, S, heads, 0)
    def fus(r):
        ops.gemm_mfma_qkv_rope(a[r % 3], w[r % 4], qkv, ops.RopeKV(kc, vc, cos, sin, B, S, heads, 0), t)
    res[t] = {"gemm+rope_kv_us": timed(unf), "fused_us": timed(fus)}
print(json.dumps(res))
