#!/usr/bin/env python3
"""4-wave tiles (97/98/99): results against the 16-wave tile (same MFMA, same K order -> identical bits expected) and
time per launch at the hot shapes (cold operands)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)
TILES = [int(x) for x in os.environ.get("P9_TILES", "9,1,97,98,99").split(",")]
for (M, N, K, epi) in [(300, 520, 128, 0), (2688, 5120, 5120, 0), (1000, 1024, 192, 1), (777, 1536, 640, 2)]:
    a = torch.randn((M, K), device=d).to(torch.bfloat16); w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=d) if epi != 2 else None
    ref = ops.gemm_mfma(a, w, bias, epilogue=epi, tile_hint=9)
    for t in TILES[2:]:
        o = ops.gemm_mfma(a, w, bias, epilogue=epi, tile_hint=t)
        o32 = ops.gemm_mfma(a, w, bias, epilogue=0, out_dtype=torch.float32, tile_hint=t) if epi == 0 else None
        torch.cuda.synchronize()
        print("check", (M, N, K, epi), "tile", t, "bit-identical" if torch.equal(o, ref) else f"DIFF max {(o.float()-ref.float()).abs().max().item()}",
              "" if o32 is None else ("f32 ok" if torch.equal(o32, ops.gemm_mfma(a, w, bias, out_dtype=torch.float32, tile_hint=9)) else "f32 DIFF"), flush=True)
for (M, N, K, epi) in [(2688, 27648, 5120, 2), (2688, 15360, 5120, 0), (2688, 5120, 13824, 0), (32768, 4096, 1024, 1), (32768, 1024, 4096, 0), (32896, 3072, 1024, 0), (8192, 8192, 8192, 0)]:
    As = [torch.randn((M, K), device=d).to(torch.bfloat16) for _ in range(3)]
    Ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.zeros(N, device=d) if epi != 2 else None
    out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
    res = {}
    for t in TILES:
        ts = []
        for r in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_mfma(As[r % 3], Ws[r % 4], bias, epilogue=epi, out=out, tile_hint=t); e1.record()
            torch.cuda.synchronize()
            if r >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
        med = statistics.median(ts)
        res[t] = (round(med, 1), round(2.0 * M * N * K / med / 1e6))
    print(json.dumps({"shape": f"{M}x{N}x{K}/e{epi}", "us,TF by tile": res}), flush=True)
    del As, Ws, out
