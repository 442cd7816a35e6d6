#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (run through gpurun).  Prints one JSON object per line:
GEMM TF/s for the hot-path shapes under every tile choice, attention and row-kernel rates.
Random (not zero) operands — zero-filled data inflates MFMA clocks (cdna_hip_programming §5.4 r25)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    d = torch.device("cuda:0")
    quick = "--quick" in sys.argv
    shapes = [
        # (name, M, N, K, epilogue)
        ("vit.qkv  F=32", 8224, 3072, 1024, 0),
        ("vit.fc1  F=32", 8224, 4096, 1024, 1),
        ("vit.fc2  F=32", 8224, 1024, 4096, 0),
        ("vit.out  F=32", 8224, 1024, 1024, 0),
        ("vit.qkv  F=128", 32896, 3072, 1024, 0),
        ("vit.fc1  F=128", 32896, 4096, 1024, 1),
        ("vit.fc2  F=128", 32896, 1024, 4096, 0),
        ("l7b.qkv  c2", 1312, 12288, 4096, 0),
        ("l7b.o    c2", 1312, 4096, 4096, 0),
        ("l7b.gu   c2", 1312, 22016, 4096, 2),
        ("l7b.down c2", 1312, 4096, 11008, 0),
        ("l13b.qkv c3", 2688, 15360, 5120, 0),
        ("l13b.gu  c3", 2688, 27648, 5120, 2),
        ("l13b.down c3", 2688, 5120, 13824, 0),
        ("square 4096", 4096, 4096, 4096, 0),
        ("square 8192", 8192, 8192, 8192, 0),
    ]
    if quick:
        shapes = shapes[:4] + shapes[7:11] + shapes[-2:-1]
    if "--gemm-only" in sys.argv:
        only_gemm = True
    else:
        only_gemm = False
    for name, M, N, K, epi in shapes:
        a = torch.randn((M, K), device=d).to(torch.bfloat16)
        w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
        bias = torch.randn((N,), device=d) if epi != 2 else None
        row = {"kernel": "gemm", "name": name, "M": M, "N": N, "K": K, "epi": epi}
        for tile in (1, 5, 51, 55, 61, 65):
            out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
            t = timeit(lambda: ops.gemm_mfma(a, w, bias, epilogue=epi, out=out, tile_hint=tile))
            row[f"tile{tile}_TF"] = round(2.0 * M * N * K / t / 1e12, 1)
        out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_mfma(a, w, bias, epilogue=epi, out=out, tile_hint=0))
        row["auto_TF"] = round(2.0 * M * N * K / t / 1e12, 1)
        for tile in ():
            t = timeit(lambda: ops.gemm_streamk(a, w, bias, epilogue=epi, out=out, tile_hint=tile))
            row[f"sk{tile}_TF"] = round(2.0 * M * N * K / t / 1e12, 1)
        print(json.dumps(row), flush=True)
        del a, w

    if only_gemm:
        return
    # ViT attention
    for F in (32, 128):
        qkv = torch.randn((F * 257, 3072), device=d).to(torch.bfloat16)
        out = torch.empty((F * 257, 1024), device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.vit_attention(qkv, F, out=out))
        print(json.dumps({"kernel": "vit_attention", "F": F, "us": round(t * 1e6, 1),
                          "TF": round(4.0 * 257 * 257 * 64 * 16 * F / t / 1e12, 1)}), flush=True)
    # Llama attention prefill
    for (B, S, heads) in ((4, 328, 32), (8, 336, 40)):
        qkv = torch.randn((B * S, 3 * heads * 128), device=d).to(torch.bfloat16)
        kc = torch.randn((B, heads, 512, 128), device=d).to(torch.bfloat16)
        vc = torch.randn((B, heads, 512, 128), device=d).to(torch.bfloat16)
        out = torch.empty((B * S, heads * 128), device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.llama_attention(qkv, kc, vc, None, B, S, heads, 0, out=out))
        print(json.dumps({"kernel": "llama_attention", "B": B, "S": S, "heads": heads, "us": round(t * 1e6, 1),
                          "TF_causal": round(2.0 * S * (S + 1) * 128 * heads * B / t / 1e12, 1)}), flush=True)
    # row kernels
    for (M, D) in ((8224, 1024), (32896, 1024), (2688, 5120)):
        x = torch.randn((M, D), device=d)
        g = torch.ones((D,), device=d)
        b = torch.zeros((D,), device=d)
        y = torch.empty((M, D), device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.layernorm(x, g, b, 1e-5, out=y))
        print(json.dumps({"kernel": "layernorm", "M": M, "D": D, "us": round(t * 1e6, 1),
                          "GBps": round(M * D * 6 / t / 1e9, 1)}), flush=True)
    # GEMV (decode) weight streaming
    for (N, K) in ((15360, 5120), (5120, 13824), (32008, 5120)):
        a = torch.randn((1, K), device=d).to(torch.bfloat16)
        w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
        out = torch.empty((1, N), device=d, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemv(a, w, out=out))
        print(json.dumps({"kernel": "gemv", "N": N, "K": K, "us": round(t * 1e6, 1),
                          "GBps": round(N * K * 2 / t / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
