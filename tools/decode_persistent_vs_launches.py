"""Persistent vs launches, step by step, with a report of WHERE the residual stream differs (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valley_amd import decode, ops
from valley_amd.llama import HipLlama

ops.GEMM_MODE = "tiles"
name, layers, B, S, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
graph = len(sys.argv) > 6 and sys.argv[6] == "graph"
kinds = {"lp": (False, True), "ll": (False, False), "pp": (True, True)}[sys.argv[7] if len(sys.argv) > 7 else "lp"]
SH = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6)}[name]
ll = HipLlama(SH["H"], SH["heads"], SH["I"], layers, 512, SH["eps"]).init_random(seed=1)
x = (torch.randn((B * S, SH["H"]), device="cuda") * 0.5)
sess = []
for p in kinds:
    decode.PERSISTENT = p
    cache = ll.new_cache(B, S + steps + 8)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    if B > 1:
        cache.key_valid[B - 1, :11] = 0
    cache.seq_len = 0
    ll.forward(x.clone(), B, S, cache)
    s = decode.DecodeSession(ll, cache, use_graph=graph)
    s.begin(torch.tensor([3, 7][:B], device="cuda"))
    sess.append(s)
bad = 0
for it in range(steps):
    for s in sess:
        s.step()
    torch.cuda.synchronize()
    a, b = sess[0].h, sess[1].h
    if not torch.equal(a, b):
        d = (a != b)
        rows = d.any(1).nonzero().flatten().tolist()
        cols = d.any(0).nonzero().flatten().tolist()
        print(f"step {it}: {int(d.sum())} of {d.numel()} differ; rows {rows}; cols {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)}); max {float((a - b).abs().max()):.3e}", flush=True)
        bad += 1
        # resynchronise the persistent session's state with the reference so that later steps are comparable
        sess[1].h.copy_(a); sess[1].tok.copy_(sess[0].tok)
        for li in range(ll.L):
            sess[1].cache.k[li].copy_(sess[0].cache.k[li]); sess[1].cache.v[li].copy_(sess[0].cache.v[li])
print("mismatching steps:", bad, "of", steps, "kinds", kinds)
