#!/usr/bin/env python3
"""Anatomy of vit_attn_kernel from s_memtime stamps (a -DVLY_VIT_TIMING=1 build of attention.hip: tools/ab_lib.py build vtime --src
attention.hip -DVLY_VIT_TIMING=1 [more flags]).  Every wave of every workgroup stamps: start, staging issued, barrier passed, and per
query tile: QK^T issued, softmax done, PV issued, stores issued; plus HW_ID / XCC_ID.  Prints per-wave medians (shader cycles) and, per CU
(s_memtime is per XCD, so only stamps of one XCD are compared), how the workgroups followed each other.

  python tools/vit_attn_times.py [variant=vtime] [frames=128]
"""
import collections
import ctypes
import os
import statistics as st
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "vtime"
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
WG, NW, N = 4096, 8, 24
buf = (ctypes.c_ulonglong * (WG * NW * N))()
assert L.vlydbg_vit_timing_read(buf) == 0
nwg = min(WG, F * 16)
rows = [[buf[(w * NW + v) * N + i] for i in range(N)] for w in range(nwg) for v in range(NW)]


def med(xs):
    return int(st.median(xs)) if xs else 0


def last(r):
    return max(x for x in r[2:N - 1] if x)


print(f"variant {name}, {F} frames ({nwg} workgroups): {us:.1f} us this launch")
print("per wave, medians over", len(rows), "waves (shader cycles):")
print("  start -> staging issued   ", med([r[2] - r[0] for r in rows]))
print("  staging issued -> barrier ", med([r[3] - r[2] for r in rows]))
for i in range(3):
    sel = [r for r in rows if r[4 + 4 * i + 3] > 0]
    if not sel:
        continue
    prev = 3 if i == 0 else 4 + 4 * (i - 1) + 3
    print(f"  tile round {i} ({len(sel)} waves): QK^T {med([r[4 + 4 * i] - r[prev] for r in sel])}  softmax {med([r[5 + 4 * i] - r[4 + 4 * i] for r in sel])}"
          f"  PV {med([r[6 + 4 * i] - r[5 + 4 * i] for r in sel])}  store {med([r[7 + 4 * i] - r[6 + 4 * i] for r in sel])}")
life = [last(r) - r[0] for r in rows]
print("  wave lifetime (start -> last stamp): median", med(life), "max", max(life))
wgs = []                                             # (xcc, se, cu, start, end, blockIdx)
for w in range(nwg):
    rr = rows[w * NW:(w + 1) * NW]
    hw = rr[0][1] & 0xffffffff
    xcc = (rr[0][1] >> 32) & 0xf
    wgs.append((xcc, (hw >> 13) & 7, (hw >> 8) & 15, min(r[0] for r in rr), max(last(r) for r in rr), w))
print("  workgroup span: median", med([e - b for *_, b, e, _ in wgs]), "min", min(e - b for *_, b, e, _ in wgs), "max", max(e - b for *_, b, e, _ in wgs))
bycu = collections.defaultdict(list)
for x in wgs:
    bycu[x[:3]].append(x[3:])
print("  XCDs seen:", sorted({k[0] for k in bycu}), " CUs seen:", len(bycu), " workgroups per CU: min", min(len(v) for v in bycu.values()), "max", max(len(v) for v in bycu.values()))
byx = collections.defaultdict(list)
for k, v in bycu.items():
    byx[k[0]] += v
spans = {x: max(e for _, e, _ in v) - min(b for b, _, _ in v) for x, v in byx.items()}
print("  per-XCD span, first start -> last stamp:", {x: spans[x] for x in sorted(spans)}, f"-> clock {st.median(spans.values()) / us / 1e3:.2f} GHz if the span is the launch")
busy, two, gaps, cu_span = [], [], [], []
for k, v in bycu.items():
    v.sort()
    t0, t1 = v[0][0], max(e for _, e, _ in v)
    ev = sorted([(b, 1) for b, _, _ in v] + [(e, -1) for _, e, _ in v])
    n, prev, t_any, t_two = 0, t0, 0, 0
    for t, dlt in ev:
        if n >= 1:
            t_any += t - prev
        if n >= 2:
            t_two += t - prev
        n += dlt
        prev = t
    busy.append(t_any / (t1 - t0))
    two.append(t_two / (t1 - t0))
    cu_span.append(t1 - t0)
    ends = sorted(e for _, e, _ in v)
    starts = sorted(b for b, _, _ in v)[2:]          # the third and later workgroups of a CU each follow an end
    gaps += [b - e for b, e in zip(starts, ends)]
print("  per CU: span median", med(cu_span), " fraction of it with >= 1 workgroup resident", round(st.median(busy), 3), " with 2 resident", round(st.median(two), 3))
print("  workgroup start minus the end it followed (k-th start vs (k-2)-th end on the CU): median", med(gaps), "p10", int(sorted(gaps)[len(gaps) // 10]), "p90", int(sorted(gaps)[len(gaps) * 9 // 10]))
k0 = sorted(bycu)[0]
print("  one CU", k0, "(start, end, blockIdx) relative:", [(b - bycu[k0][0][0], e - bycu[k0][0][0], i) for b, e, i in bycu[k0]])
