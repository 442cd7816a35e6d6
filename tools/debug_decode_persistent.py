"""One eager persistent decode step on a 2-layer model at 13B shapes; dumps the grid-barrier words (debugging aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valley_amd import decode, ops, weights as W
from valley_amd.llama import HipLlama

ops.GEMM_MODE = "tiles"
name = sys.argv[1] if len(sys.argv) > 1 else "13b"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SH = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6)}[name]
ll = HipLlama(SH["H"], SH["heads"], SH["I"], layers, 512, SH["eps"]).init_random(seed=1)
B, S = 1, 328
cache = ll.new_cache(B, S + 40)
cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
x = (torch.randn((B * S, SH["H"]), device="cuda") * 0.5)
cache.seq_len = 0
ll.forward(x, B, S, cache)
decode.PERSISTENT = True
use_graph = (sys.argv[3] == "graph") if len(sys.argv) > 3 else False
sess = decode.DecodeSession(ll, cache, use_graph=use_graph)
print("persistent:", sess.persistent, flush=True)
sess.begin(torch.tensor([3], device="cuda"))
for it in range(3):
    t0 = time.time()
    sess.step()
    torch.cuda.synchronize()
    s = sess.sync.cpu().numpy().astype("uint32")
    print(f"step {it}: {1e3 * (time.time() - t0):.2f} ms  cnt", s[0:128:16].tolist(), "top", int(s[128]), "gen", s[144:272:16].tolist(), "abort", int(s[272]),
          "h finite", bool(torch.isfinite(sess.h).all()), flush=True)
