#!/usr/bin/env python3
"""Round 6: the persistent GEMM on the 32x32x16 MFMA (tile hint 397, gemm_p32.hip) against fp32 PyTorch and against the 16x16x32
persistent kernel (hints 197-199): correctness on the path's shapes and their edges, then interleaved timing on cold operands.
p32_check.py [check] [time] [reps]"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops, runtime  # noqa: E402

d = torch.device("cuda:0")
HINTS = [int(x) for x in os.environ.get("P32_HINTS", "397,398").split(",")]
HALF = runtime.HALF


def ref(a, w, bias, epi):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    elif epi == 3:
        y = torch.relu(y)
    elif epi == 2:
        y = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    return y


def check():
    torch.manual_seed(0)
    bad = 0
    cases = [(512, 512, 640, 0, True, False), (256, 256, 576, 0, False, False), (700, 1024, 1024, 0, True, False),
             (2688, 5120, 5120, 0, False, True), (2688, 2048, 1024, 2, False, True), (1000, 1032, 704, 0, True, False),
             (2056, 4096, 1024, 1, True, False), (2056, 1024, 4096, 0, True, True), (3000, 3072, 1024, 3, True, False),
             (336, 32008, 1024, 0, False, False), (8224, 3072, 1024, 0, True, False), (2688, 27648, 5120, 2, False, True)]
    for M, N, K, epi, has_bias, packed in cases:
        a = torch.randn((M, K), device=d).to(HALF)
        w = (torch.randn((N, K), device=d) * 0.05).to(HALF)
        bias = torch.randn(N, device=d) if has_bias else None
        wp = ops.PackedWeight(w) if packed else w
        o2 = ops.gemm_mfma(a, wp, bias, epilogue=epi, tile_hint=197)
        r = ref(a, w, bias, epi)
        err2 = (o2.float() - r).abs().max().item()
        scale = r.abs().max().item()
        row = {"shape": f"{M}x{N}x{K}/e{epi}", "bias": has_bias, "packed": packed, "err_197": round(err2, 5), "max": round(scale, 3)}
        for hint in HINTS:
            out = ops.gemm_mfma(a, wp, bias, epilogue=epi, tile_hint=hint)
            torch.cuda.synchronize()
            err = (out.float() - r).abs().max().item()
            ok = err <= max(2.5 * err2, 1e-2 * scale) and torch.isfinite(out.float()).all().item()
            bad += not ok
            row[f"err_{hint}"] = round(err, 5)
            row[f"ok_{hint}"] = bool(ok)
        print(json.dumps(row), flush=True)
    return bad


def time_shapes(reps):
    shapes = [(2688, 27648, 5120, 2, False, "13B gate|up"), (2688, 5120, 13824, 0, False, "13B down"), (2688, 15360, 5120, 0, False, "13B qkv (no rope)"),
              (2688, 5120, 5120, 0, False, "13B o"), (32896, 4096, 1024, 1, True, "ViT fc1"), (32896, 1024, 4096, 0, True, "ViT fc2"),
              (32896, 3072, 1024, 0, True, "ViT qkv"), (32896, 1024, 1024, 0, True, "ViT out")]
    for M, N, K, epi, has_bias, name in shapes:
        As = [torch.randn((M, K), device=d).to(HALF) for _ in range(3)]
        Ws = [ops.PackedWeight((torch.randn((N, K), device=d) * 0.05).to(HALF)) for _ in range(3)]
        bias = torch.randn(N, device=d) if has_bias else None
        out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=HALF)
        tiles = HINTS + [197, 198, 199]
        ts = {t: [] for t in tiles}
        for r in range(reps + 2):
            for t in tiles:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm_mfma(As[r % 3], Ws[r % 3], bias, epilogue=epi, out=out, tile_hint=t)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    ts[t].append(e0.elapsed_time(e1) * 1e3)
        row = {"gemm": name, "shape": f"{M}x{N}x{K}/e{epi}"}
        for t in tiles:
            med = statistics.median(ts[t])
            row[f"us_{t}"] = round(med, 1)
            row[f"TF_{t}"] = round(2.0 * M * N * K / med / 1e6, 1)
        print(json.dumps(row), flush=True)
        del As, Ws


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
        print("check:", "FAILED" if rc else "ok", flush=True)
    if "time" in what:
        reps = int(what[-1]) if what[-1].isdigit() else 10
        time_shapes(reps)
    sys.exit(1 if rc else 0)
