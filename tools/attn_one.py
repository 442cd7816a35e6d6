#!/usr/bin/env python3
"""Run ONE attention configuration a few times (for rocprofv3 --pmc): attn_one.py vit F | llama B S heads"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops  # noqa: E402

d = torch.device("cuda:0")
kind = sys.argv[1]
if kind == "vit":
    F = int(sys.argv[2])
    qkv = torch.randn((F * 257, 3072), device=d).to(torch.bfloat16)
    for _ in range(8):
        ops.vit_attention(qkv, F)
else:
    B, S, heads = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    qkv = torch.randn((B * S, 3 * heads * 128), device=d).to(torch.bfloat16)
    kc = torch.zeros((B, heads, S, 128), device=d, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    cos = torch.ones((S, 64), device=d)
    sin = torch.zeros((S, 64), device=d)
    for _ in range(8):
        ops.rope_kv(qkv, kc, vc, cos, sin, B, S, heads, 0)
        ops.llama_attention(qkv, kc, vc, None, B, S, heads, 0)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.llama_attention(qkv, kc, vc, None, B, S, heads, 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"llama_attention B={B} S={S} heads={heads}: median {sorted(ts)[len(ts) // 2]:.1f} us, min {min(ts):.1f} us")
torch.cuda.synchronize()
