#!/usr/bin/env python3
"""Time ONE GEMM configuration in its own process (for settings read once per process, e.g. VLY_TILE_GM):
gemm_time.py M N K epi tile [reps] -> one JSON line (median us, TFLOP/s).  Activations rotate through 3 copies and
weights through 4, so every launch reads operands that left the caches."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

M, N, K, epi, tile = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
d = torch.device("cuda:0")
As = [torch.randn((M, K), device=d).to(torch.bfloat16) for _ in range(3)]
Ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(4)]
bias = torch.zeros(N, device=d)
out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
ts = []
for r in range(reps + 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_mfma(As[r % 3], Ws[r % 4], bias if epi != 2 else None, epilogue=epi, out=out, tile_hint=tile)
    e1.record()
    torch.cuda.synchronize()
    if r >= 3:
        ts.append(e0.elapsed_time(e1) * 1e3)
med = statistics.median(ts)
print(json.dumps({"shape": f"{M}x{N}x{K}/e{epi}", "tile": tile, "gm": os.environ.get("VLY_TILE_GM", "auto"), "us": round(med, 1),
                  "TFLOPs": round(2.0 * M * N * K / med / 1e6, 1)}), flush=True)
