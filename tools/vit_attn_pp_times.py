#!/usr/bin/env python3
"""Anatomy of vit_attn_pp_kernel (two persistent 8-wave workgroups per CU, attention_vit_pp.inc) from s_memtime stamps:
  python tools/ab_lib.py build pptime --src attention.hip -DVLY_VIT_TIMING=1 -DVLY_VIT_PP=1
  python tools/vit_attn_pp_times.py [variant=pptime] [frames=128]
"""
import ctypes
import os
import statistics as st
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "pptime"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = ctypes.CDLL(os.path.join(ROOT, "valley_amd", "lib", "variants", f"libvalley_hip_{name}.so"))
P, I = ctypes.c_void_p, ctypes.c_int
L.vly_vit_attention.restype = I
L.vly_vit_attention.argtypes = [P, P, I, P]
d = torch.device("cuda:0")
qkvs = [torch.randn((F * 257, 3072), device=d).to(torch.bfloat16) for _ in range(3)]
out = torch.empty((F * 257, 1024), device=d, dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
for i in range(5):
    assert L.vly_vit_attention(qkvs[i % 3].data_ptr(), out.data_ptr(), F, s) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.vly_vit_attention(qkvs[2].data_ptr(), out.data_ptr(), F, s) == 0
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
buf = (ctypes.c_ulonglong * (4096 * 8 * 24))()
assert L.vlydbg_vit_timing_read(buf) == 0
NW, N = 8, 32
allrows = [[buf[(w * NW + v) * N + i] for i in range(N)] for w in range(512) for v in range(NW)]
rows = [r for r in allrows if r[0] and r[16 + 12]]
print(f"variant {name} (vit_attn_pp_kernel), {F} frames: {us:.1f} us this launch; {len(rows)} stamped waves")
names = ["wait K pieces", "B1", "QK1 (+4 V pieces)", "softmax1", "wait V pieces", "B2", "PV1+store", "partial", "QK2", "B3", "softmax2 (+3 K pieces)", "PV2 (+2 K pieces, Q loads)+store"]


def med(xs):
    return int(st.median(xs))


for hd in range(2):
    b = 16 * hd
    print(f"  head {hd + 1} of the walk:", "  ".join(f"{n} {med([r[b + i + 1] - r[b + i] for r in rows])}" for i, n in enumerate(names)), " | total", med([r[b + 12] - r[b] for r in rows]))
print("  head period (start to next start): ", med([r[16] - r[0] for r in rows]))
print("  per wave, head 2 (segments in the order above):")
for v in range(NW):
    wv = [allrows[w * NW + v] for w in range(512) if allrows[w * NW + v][0] and allrows[w * NW + v][28]]
    print(f"    wave {v}: ", " ".join(f"{med([r[16 + i + 1] - r[16 + i] for r in wv]):5d}" for i in range(12)))
