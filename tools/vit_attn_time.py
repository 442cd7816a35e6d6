#!/usr/bin/env python3
"""vly_vit_attention: time per launch at F frames and the error against an fp32 torch evaluation of the same bf16 inputs.
VLY_VIT_ATTN=1 / 4 forces the workgroup-per-head / the persistent kernel (read once per process; default: by frame count): run once per
setting.  One JSON line."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops, runtime  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(5)
    qkvs = [(torch.randn((F * 257, 3072), generator=g, device=d) * 1.5).to(runtime.HALF) for _ in range(3)]
    out = torch.empty((F * 257, 1024), dtype=runtime.HALF, device=d)
    ops.vit_attention(qkvs[0], F, out=out)
    torch.cuda.synchronize()
    # fp32 reference over every frame (in slabs of 8: the fp32 scores of 128 frames would be 0.5 GB)
    num = den = mx = 0.0
    for f0 in range(0, F, 8):
        n = min(8, F - f0)
        x = qkvs[0][f0 * 257:(f0 + n) * 257].float().view(n, 257, 3, 16, 64)
        q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
        ref = (p @ v).transpose(1, 2).reshape(n * 257, 1024)
        got = out[f0 * 257:(f0 + n) * 257].float()
        num += float((got - ref).pow(2).sum())
        den += float(ref.pow(2).sum())
        mx = max(mx, float((got - ref).abs().max()))
    rel = (num / den) ** 0.5
    ts = []
    for i in range(43):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.vit_attention(qkvs[i % 3], F, out=out)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"vit_attn_frames": F, "kernel": os.environ.get("VLY_VIT_ATTN", "default"), "median_us": round(statistics.median(ts), 1),
                      "min_us": round(min(ts), 1), "rel_l2_vs_fp32": round(rel, 5), "max_abs": round(mx, 4)}), flush=True)


if __name__ == "__main__":
    main()
