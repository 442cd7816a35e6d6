#!/usr/bin/env python3
"""Information only (not a product path): the vendor library's bf16 GEMM (torch F.linear -> hipBLASLt/rocBLAS) beside
vly_gemm_bf16 at the hot shapes, same process, same cold-operand rotation.  Answers "how far is the hand-written kernel from
what the library reaches on this box" — the ceiling the roofline fraction should be read against."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

d = torch.device("cuda:0")
SHAPES = [(2688, 27648, 5120), (2688, 15360, 5120), (2688, 5120, 13824), (2688, 5120, 5120), (32768, 4096, 1024),
          (32768, 1024, 4096), (32896, 3072, 1024), (32768, 1024, 1024), (8192, 8192, 8192), (4096, 4096, 4096)]


def timed(fn, reps=20):
    ts = []
    for r in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(r)
        e1.record()
        torch.cuda.synchronize()
        if r >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


for M, N, K in SHAPES:
    As = [torch.randn((M, K), device=d).to(torch.bfloat16) for _ in range(3)]
    Ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(4)]
    out = torch.empty((M, N), device=d, dtype=torch.bfloat16)
    t_lib = timed(lambda r: torch.nn.functional.linear(As[r % 3], Ws[r % 4], out=None))
    t_lib_out = timed(lambda r: torch.mm(As[r % 3], Ws[r % 4].t(), out=out))
    t_vly = float("nan")
    if os.environ.get("PROBE_LIB_ONLY") != "1":
        for r in range(40):                                  # let the online tuner settle
            ops.gemm(As[r % 3], Ws[r % 4], out=out)
        t_vly = timed(lambda r: ops.gemm(As[r % 3], Ws[r % 4], out=out))
    fl = 2.0 * M * N * K / 1e6
    print(json.dumps({"shape": f"{M}x{N}x{K}", "lib_us": round(min(t_lib, t_lib_out), 1), "lib_TF": round(fl / min(t_lib, t_lib_out), 1),
                      "vly_us": round(t_vly, 1), "vly_TF": round(fl / t_vly, 1)}), flush=True)
    del As, Ws, out
