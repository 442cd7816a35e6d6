#!/usr/bin/env python3
"""How far ahead of the GPU does the host run?  Enqueue time of c2 steps (no sync) vs their GPU time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from valley_amd import valley_model as vm, weights as W  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda:0")
B, T, H, I, L = cfg["B"], cfg["T"], cfg["H"], cfg["I"], cfg["L"]
S = 320 + T
config = vm.ValleyConfig(vocab_size=bench.VOCAB_TEXT + 6, hidden_size=H, intermediate_size=I, num_hidden_layers=L,
                         num_attention_heads=cfg["heads"], num_key_value_heads=cfg["heads"], rms_norm_eps=cfg["eps"],
                         max_position_embeddings=2048)
config.use_mm_proj, config.mm_hidden_size, config.mm_vision_select_layer = True, 1024, -2
model = vm.ValleyLlamaForCausalLM(config, device=dev)
mm = model.get_model()
mm.llama.init_random(seed=0)
tower = vm.build_vision_tower(None, device=dev)
tower.init_random(seed=0, layers=23)
for k, v in W.SPECIAL_IDS(bench.VOCAB_TEXT).items():
    setattr(tower.config, k, v)
mm.initialize_vision_modules(tower, -2)
frames = torch.randn((B, T, 3, 224, 224), device=dev).to(torch.bfloat16)
ids = torch.from_numpy(W.synthetic_prompt(7, T, bench.VOCAB_TEXT)).view(1, S).repeat(B, 1)
cache = mm.llama.new_cache(B, S)


def step():
    pooled, _ = mm.encode_clips(frames)
    visual = mm.project_pooled(pooled)
    cache.seq_len = 0
    return model(input_ids=ids, past_key_values=cache, use_cache=True, visual_tokens=visual, frames_per_clip=[T] * B)


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step "
      f"(GPU-bound if enqueue << total)")

if os.environ.get("HOST_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
