#!/usr/bin/env python3
"""Sweep tile hints over GEMM shapes with cold caches: gemm_sweep.py "M,N,K,epi;..." "tile,tile,..." ["sk tile,tile,..."]
Prints one JSON line per (shape, tile): TFLOP/s of the median launch and max |err| vs a torch fp32 GEMM."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1].split(";")]
tiles = [("tile", int(t)) for t in sys.argv[2].split(",") if t]
if len(sys.argv) > 3:
    tiles += [("sk", int(t)) for t in sys.argv[3].split(",") if t]
WARM_W = bool(int(os.environ.get("SWEEP_WARM_W", "0")))
d = torch.device("cuda:0")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)
for M, N, K, epi in shapes:
    a = torch.randn((M, K), device=d).to(torch.bfloat16)
    w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
    ref = a[:256].float() @ w.float().t()
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 2:
        g, u = ref[:, 0::2], ref[:, 1::2]
        ref = torch.nn.functional.silu(g) * u
    out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
    warm = torch.empty_like(a)
    warm_w = torch.empty_like(w) if WARM_W else None
    for kind, t in tiles:
        fn = ops.gemm_mfma if kind == "tile" else ops.gemm_streamk
        try:
            ts = []
            for _ in range(7):
                flush.zero_()
                warm.copy_(a)                              # A is produced just before the GEMM: cache-warm
                if WARM_W:
                    warm_w.copy_(w)                        # experiment: weights resident in the Infinity Cache
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(a, w, epilogue=epi, out=out, tile_hint=t)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            ms = ts[len(ts) // 2]
            err = (out[:256].float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            print(json.dumps({"M": M, "N": N, "K": K, "epi": epi, "tile": t if kind == "tile" else f"sk{t}", "ms": round(ms, 4),
                              "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "rel_err": round(err, 5)}), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"M": M, "N": N, "K": K, "tile": f"{kind}{t}", "error": str(ex)[:200]}), flush=True)
