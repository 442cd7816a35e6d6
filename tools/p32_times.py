#!/usr/bin/env python3
"""Anatomy of the 32x32x16 persistent GEMM's tile loop from its own s_memtime stamps (variant built with -DVLY_P32_TIMING=1:
tools/ab_lib.py build NAME --src gemm_p32.hip -DVLY_P32_TIMING=1).   p32_times.py NAME M,N,K,epi [...]
Stamps per tile: start | after the HEAD K tiles (parked stores leave there) | after the last K tile | after the drain."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARDIR = os.path.join(ROOT, "valley_amd", "lib", "variants")
P, I = ctypes.c_void_p, ctypes.c_int
d = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
HEAD = int(os.environ.get("P32_HEAD", "8"))
HINT = int(os.environ.get("P32_HINT", "397"))
for n in sys.argv[1].split(","):
    L = ctypes.CDLL(os.path.join(VARDIR, f"libvalley_hip_{n}.so"))
    L.vly_gemm_bf16.restype = I
    L.vly_gemm_bf16.argtypes = [P] * 5 + [I] * 10 + [P]
    for sh in sys.argv[2:]:
        M, N, K, epi = (int(x) for x in sh.split(","))
        a = torch.randn((M, K), device=d).to(torch.bfloat16)
        w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=d) if epi != 2 else None
        out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
        for _ in range(3):
            assert L.vly_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, None, out.data_ptr(), M, N, K, K, K,
                                   out.shape[1], 0, epi, 0, HINT, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):                                  # back to back: settled (power-limited) clock
            L.vly_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, None, out.data_ptr(), M, N, K, K, K,
                            out.shape[1], 0, epi, 0, HINT, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        host = (ctypes.c_ulonglong * (64 * 65))()
        assert L.vlydbg_p32_timing_read(host) == 0
        nk = K // 64
        head, rest, drain, total = [], [], [], []
        for wg in range(64):
            cnt = min(int(host[wg * 65]), 64)
            ts = [int(host[wg * 65 + 1 + q]) for q in range(cnt)]
            if cnt < 5:
                continue
            total.append(ts[-1] - ts[0])
            for t in range((cnt - 1) // 4):
                s0, s1, s2, s3 = ts[4 * t:4 * t + 4]
                if t > 0:                                   # (the first tile of a workgroup carries the cold start)
                    head.append((s1 - s0) / HEAD)
                    rest.append((s2 - s1) / (nk - HEAD))
                    drain.append(s3 - s2)
        if not head:
            print(f"{n} {sh}: one tile per workgroup")
            continue
        tile = HEAD * statistics.median(head) + (nk - HEAD) * statistics.median(rest) + statistics.median(drain)
        print(f"{n:8s}/{HINT} {sh}: kernel {statistics.median(total):.0f} clk | per K tile: head {statistics.median(head):.0f}  rest {statistics.median(rest):.0f} "
              f"(ideal 2048) | drain {statistics.median(drain):.0f} | tile {tile:.0f} = {2048 * nk / tile:.3f} of MFMA-bound | {us:.1f} us = "
              f"{2.0 * M * N * K / us / 1e6:.0f} TFLOP/s, clock {statistics.median(total) / us / 1e3:.2f} GHz", flush=True)
