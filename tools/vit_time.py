#!/usr/bin/env python3
"""ViT-L/14 tower time for F frames (default 128, the c3 batch) under the current environment (VALLEY_VIT_CHUNK,
VALLEY_ROW_SPLIT*, ...): frames/s and fraction of the bf16 peak.  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import ops, valley_model as vm  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d = torch.device("cuda:0")
    tower = vm.build_vision_tower(None, device=d)
    tower.init_random(seed=0, layers=23)
    frames = torch.randn((F, 3, 224, 224), device=d).to(torch.bfloat16)
    n = 0
    while True:                                         # let the online tuner settle
        tower.encode(frames, -2)
        torch.cuda.synchronize()
        n += 1
        if ops.tuning_pending() == 0 or n > 300:
            break
    for _ in range(3):
        tower.encode(frames, -2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 10
    for _ in range(it):
        tower.encode(frames, -2)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / it
    tf = F * 155.29e9 / t / 1e12
    print(json.dumps({"vit_frames": F, "ms": round(t * 1e3, 3), "frames_per_s": round(F / t, 1), "TFLOPs": round(tf, 1),
                      "frac_bf16_peak": round(tf / 2500, 4), "tune_passes": n,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("VALLEY_")}}), flush=True)


if __name__ == "__main__":
    main()
