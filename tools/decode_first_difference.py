"""1-layer model: launches vs persistent; at the first mismatching step report which intermediate differs first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valley_amd import decode, ops, runtime
from valley_amd.llama import HipLlama

ops.GEMM_MODE = "tiles"
name, steps = sys.argv[1], int(sys.argv[2])
SH = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6)}[name]
torch.manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
ll = HipLlama(SH["H"], SH["heads"], SH["I"], 1, 512, SH["eps"]).init_random(seed=1)
B, S = 1, 328
x = (torch.randn((B * S, SH["H"]), device="cuda") * 0.5)
sess = []
for p in (False, True):
    decode.PERSISTENT = p
    cache = ll.new_cache(B, S + steps + 8)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    cache.seq_len = 0
    ll.forward(x.clone(), B, S, cache)
    s = decode.DecodeSession(ll, cache, use_graph=False)
    s.begin(torch.tensor([3], device="cuda"))
    sess.append(s)
def cmp(name, a, b):
    a, b = a.float().flatten(), b.float().flatten()
    d = (a != b).nonzero().flatten()
    print(f"   {name}: {d.numel()} of {a.numel()} differ", ("first idx %s: %r vs %r" % (d[:6].tolist(), a[d[:3]].tolist(), b[d[:3]].tolist())) if d.numel() else "", flush=True)
bad = 0
for it in range(steps):
    h_in = []
    for s in sess:
        s.step()
    torch.cuda.synchronize()
    a, b = sess[0], sess[1]
    if not torch.equal(a.h, b.h):
        bad += 1
        print(f"step {it} (pos {S + it}): h differs, max {float((a.h - b.h).abs().max()):.3e}", flush=True)
        cmp("qkv", a.qkv, b.qkv)
        cmp("partials", a.partials, b.partials)
        cmp("mlp", a.mlp, b.mlp32.to(runtime.HALF))
        cmp("kcache", a.cache.k[0], b.cache.k[0])
        cmp("vcache", a.cache.v[0], b.cache.v[0])
        if bad >= 3:
            break
        b.h.copy_(a.h); b.tok.copy_(a.tok); b.cache.k[0].copy_(a.cache.k[0]); b.cache.v[0].copy_(a.cache.v[0])
print("mismatching steps:", bad)
