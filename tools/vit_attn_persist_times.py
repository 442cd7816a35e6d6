#!/usr/bin/env python3
"""Anatomy of vit_attn_persist_kernel (the persistent LDS-DMA ViT attention, attention_vit_persist.inc) from s_memtime stamps:
  python tools/ab_lib.py build vptime --src attention.hip -DVLY_VIT_TIMING=1
  python tools/vit_attn_persist_times.py [variant=vptime] [frames=128]
Per wave and head: nine stamps (head start, own pieces landed, barrier passed, first piece issued, second piece issued, QK^T issued, softmax
done, PV issued, stores + the 257th query's partial done)."""
import ctypes
import os
import statistics as st
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "vptime"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
os.environ["VLY_VIT_ATTN"] = "4"
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
NW, N = 16, 32
rows = [[buf[(w * NW + v) * N + i] for i in range(N)] for w in range(256) for v in range(NW)]
rows = [r for r in rows if r[0] and r[28]]
print(f"variant {name} (vit_attn_persist_kernel), {F} frames: {us:.1f} us this launch; {len(rows)} stamped waves")
names = ["wait own pieces", "barrier", "merge (wave 0) + piece", "piece", "QK^T", "softmax", "PV", "store + 257th query partial"]


def med(xs):
    return int(st.median(xs))


for hd in range(3):
    b = 10 * hd
    print(f"  head {hd + 1} of the walk:", "  ".join(f"{n} {med([r[b + i + 1] - r[b + i] for r in rows])}" for i, n in enumerate(names)),
          " | total", med([r[b + 8] - r[b] for r in rows]))
print("  head period (start to next start): ", med([r[10] - r[0] for r in rows]), med([r[20] - r[10] for r in rows]))
print("  per wave, head 2 (segments in the order above, then the wave's arrival at the NEXT barrier relative to the workgroup's first arrival):")
allrows = [[buf[(w * NW + v) * N + i] for i in range(N)] for w in range(256) for v in range(NW)]
for v in range(NW):
    wv = [allrows[w * NW + v] for w in range(256) if allrows[w * NW + v][0] and allrows[w * NW + v][28]]
    arr = [allrows[w * NW + v][21] - min(allrows[w * NW + u][21] for u in range(NW)) for w in range(256) if allrows[w * NW][0] and allrows[w * NW][28]]
    print(f"    wave {v:2d}: ", " ".join(f"{med([r[10 + i + 1] - r[10 + i] for r in wv]):5d}" for i in range(8)), f"  arrives +{med(arr)}")
