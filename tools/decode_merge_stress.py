"""Stress of the ticket hand-off in vly_decode_attention_merged: two sessions on identical caches — merge in the attention
launch ("attn") against merge in the o GEMV's prologue ("oproj") — stepped side by side for many tokens under hipGraph replay;
every step's residual stream, logits and token must be bit-identical, and the ticket counters back at zero.
usage: decode_merge_stress.py 13b|7b <layers> <B> <S> <steps>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from valley_amd import decode, ops  # noqa: E402
from valley_amd.llama import HipLlama  # noqa: E402

ops.GEMM_MODE = "tiles"
name, layers, B, S, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
SH = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6)}[name]
ll = HipLlama(SH["H"], SH["heads"], SH["I"], layers, 512, SH["eps"]).init_random(seed=1)
x = (torch.randn((B * S, SH["H"]), device="cuda") * 0.5)
sess = []
for mode in ("oproj", "attn"):
    decode.MERGE_IN = mode
    cache = ll.new_cache(B, S + steps + 8)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    if B > 1:
        cache.key_valid[B - 1, :11] = 0
    cache.seq_len = 0
    ll.forward(x.clone(), B, S, cache)
    s = decode.DecodeSession(ll, cache, use_graph=True)
    s.begin(torch.tensor([3, 7, 11, 13, 17, 19, 23, 29][:B], device="cuda"))
    sess.append(s)
bad = 0
for it in range(steps):
    decode.MERGE_IN = "oproj"
    sess[0].step()
    decode.MERGE_IN = "attn"
    sess[1].step()
    if it % 64 == 63 or it == steps - 1:          # compare in batches: a sync per step would hide a race behind the idle GPU
        torch.cuda.synchronize()
        if not (torch.equal(sess[0].h, sess[1].h) and torch.equal(sess[0].logits, sess[1].logits) and torch.equal(sess[0].tok, sess[1].tok)):
            bad += 1
            print(f"step {it}: differs (max {float((sess[0].h - sess[1].h).abs().max()):.3e})", flush=True)
            break
torch.cuda.synchronize()
same_cache = all(torch.equal(sess[0].cache.k[li], sess[1].cache.k[li]) and torch.equal(sess[0].cache.v[li], sess[1].cache.v[li])
                 for li in range(ll.L))
print(f"{name} x {layers} layers, B = {B}, prefix {S}, {steps} steps: mismatching checkpoints {bad}, caches identical {same_cache}, "
      f"tickets left {int(sess[1].arrivals.abs().sum().item())}")
