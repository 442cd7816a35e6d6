#!/usr/bin/env python3
"""Anatomy of the persistent GEMM's tile loop from its own s_memtime stamps (builds with -DVLY_P4_TIMING=1, tools/ab_lib.py build).
  p4_boundary_times.py NAME[,NAME2] M,N,K,epi,tile [...]
Per library and shape: median cycles of the K-loop segments and of the boundary / epilogue segments over the first 64 workgroups."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valley_amd import build as b  # noqa: E402

VARDIR = os.path.join(ROOT, "valley_amd", "lib", "variants")
P, I = ctypes.c_void_p, ctypes.c_int
names = sys.argv[1].split(",")
d = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for sh in sys.argv[2:]:
    M, N, K, epi, tile = (int(x) for x in sh.split(","))
    a = torch.randn((M, K), device=d).to(torch.bfloat16)
    w = (torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16)
    out = torch.empty((M, N // 2 if epi == 2 else N), device=d, dtype=torch.bfloat16)
    for n in names:
        L = ctypes.CDLL(os.path.join(VARDIR, f"libvalley_hip_{n}.so"))
        L.vly_gemm_bf16.restype = I
        L.vly_gemm_bf16.argtypes = [P] * 5 + [I] * 10 + [P]
        for _ in range(3):
            assert L.vly_gemm_bf16(a.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), M, N, K, K, K, out.shape[1], 0, epi, 0, tile, st) == 0
        torch.cuda.synchronize()
        host = (ctypes.c_ulonglong * (64 * 65))()
        assert L.vlydbg_p4_timing_read(host) == 0
        segs = {}
        total = []
        for wg in range(64):
            cnt = min(int(host[wg * 65]), 64)
            ts = [int(host[wg * 65 + 1 + q]) for q in range(cnt)]
            if cnt < 4:
                continue
            total.append(ts[-1] - ts[0])
            for q in range(1, cnt):
                segs.setdefault(q, []).append(ts[q] - ts[q - 1])
        print(f"{n:8s} {sh}: stamps {cnt}, whole kernel {statistics.median(total):.0f} clk; segments (median clk over 64 workgroups):")
        print("   " + "  ".join(f"{q}:{statistics.median(v):.0f}" for q, v in sorted(segs.items())))
