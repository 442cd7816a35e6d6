"""Phases 1-3 of one layer (norm + q|k|v, attention, merge + o + residual): persistent kernel stopped after its third barrier
(VLY_DL_STOP=3) vs the three launches, on random residual streams.  Is the h that phase 3 WRITES already different?"""
import os, sys
os.environ["VLY_DL_STOP"] = "3"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from valley_amd import decode, ops, runtime
from valley_amd.llama import HipLlama

ops.GEMM_MODE = "tiles"
iters = int(sys.argv[1])
SH = dict(H=5120, heads=40, I=13824, eps=1e-6)
ll = HipLlama(SH["H"], SH["heads"], SH["I"], 1, 512, SH["eps"]).init_random(seed=1)
B, S = 1, 328
caches = []
for _ in range(2):
    cache = ll.new_cache(B, S + 40)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    cache.seq_len = 0
    torch.manual_seed(5)
    ll.forward((torch.randn((B * S, SH["H"]), device="cuda") * 0.5), B, S, cache)
    caches.append(cache)
decode.PERSISTENT = True
sp = decode.DecodeSession(ll, caches[1], use_graph=False)
assert sp.persistent
sp.pos.fill_(S)
L = ll.layers[0]
c0 = caches[0]
pos0 = torch.full((1,), S, dtype=torch.int32, device="cuda")
h0 = torch.empty((B, SH["H"]), dtype=torch.float32, device="cuda")
qkv0 = torch.empty((B, 3 * SH["H"]), dtype=runtime.HALF, device="cuda")
part0 = ops.decode_partials(B, ll.heads, "cuda")
table = ops.decode_layer_table(ll.layers, caches[1].k, caches[1].v, ll.device)
bad = 0
for it in range(iters):
    hin = torch.randn((B, SH["H"]), device="cuda")
    h0.copy_(hin)
    ops.gemv_rmsnorm(h0, L["ln1"], ll.eps, L["w_qkv"], out=qkv0)
    ops.decode_attention_split(qkv0, c0.k[0], c0.v[0], ll.cos, ll.sin, c0.key_valid, B, ll.heads, 0, part0, past_dev=pos0)
    ops.gemv_attnmerge(part0, L["w_o"], residual=h0, out=h0)
    sp.h.copy_(hin)
    ops.decode_layers(table, sp.h, sp.qkv, sp.partials, sp.mlp32, ll.cos, ll.sin, caches[1].key_valid, sp.pos, False, ll.heads, ll.I, ll.eps,
                      caches[1].ctx_max, sp.sync)
    torch.cuda.synchronize()
    if not torch.equal(h0, sp.h):
        d = (h0 != sp.h).flatten().nonzero().flatten()
        bad += 1
        if bad <= 5:
            print(f"iter {it}: {d.numel()} elements of h differ after phase 3: idx {d[:8].tolist()} launches {h0.flatten()[d[:4]].tolist()} persistent {sp.h.flatten()[d[:4]].tolist()} "
                  f"h_in {hin.flatten()[d[:4]].tolist()}; qkv equal {torch.equal(qkv0, sp.qkv)} partials equal {torch.equal(part0, sp.partials)}", flush=True)
print("mismatches after phase 3:", bad, "of", iters)
