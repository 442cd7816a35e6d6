#!/usr/bin/env python3
"""Stage sweeps on one MI355X: ViT-L/14 encode frames/s vs frames in flight, and Llama prefill tokens/s
vs batch, with their fraction of the 2.5 PFLOP/s bf16 peak (algorithmic FLOPs of SURVEY.md §8d)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valley_amd import valley_model as vm  # noqa: E402
from valley_amd.llama import HipLlama  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    d = torch.device("cuda:0")
    tower = vm.build_vision_tower(None, device=d)
    tower.init_random(seed=0, layers=23)
    for F in (8, 32, 64, 128, 256, 512):
        frames = torch.randn((F, 3, 224, 224), device=d).to(torch.bfloat16)
        t = timeit(lambda: tower.encode(frames, -2, chunk=256))
        tf = F * 155.29e9 / t / 1e12
        print(json.dumps({"stage": "vit_encode", "frames": F, "frames_per_s": round(F / t, 1), "TFLOPs": round(tf, 1),
                          "frac_bf16_peak": round(tf / 2500, 4)}), flush=True)
    del tower
    for name, (H, heads, I, L, eps) in {"7b": (4096, 32, 11008, 32, 1e-5), "13b": (5120, 40, 13824, 40, 1e-6)}.items():
        ll = HipLlama(H, heads, I, L, 32006, eps, device=d).init_random(seed=0)
        for B in (1, 4, 8, 16, 32):
            S = 336
            h = torch.randn((B * S, H), device=d) * 0.02
            cache = ll.new_cache(B, S)

            def run():
                cache.seq_len = 0
                x = ll.forward(h.clone(), B, S, cache)
                return ll.logits(x)
            t = timeit(run, iters=3, warm=2)
            fl = B * (S * (L * (8 * H * H + 6 * H * I) + 2 * H * 32006) + L * 2 * S * (S + 1) * H)
            print(json.dumps({"stage": "prefill", "model": name, "B": B, "S": S, "tokens_per_s": round(B * S / t, 1),
                              "TFLOPs": round(fl / t / 1e12, 1), "frac_bf16_peak": round(fl / t / 1e12 / 2500, 4)}), flush=True)
        del ll
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
