"""Phase anatomy of the persistent decode step (vly_decode_layers): per layer and phase, from s_memrealtime stamps (100 MHz)
every workgroup leaves through the library's debugging hook vlydbg_decode_timing — 0 phase start (previous barrier released),
1 activation set-up done, 2 unit loop done, 3 workgroup arrived (stores drained), 4 barrier released.
usage: decode_phase_times.py [13b|7b] [layers]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from valley_amd import decode, lib, ops
from valley_amd.llama import HipLlama

ops.GEMM_MODE = "tiles"
name = sys.argv[1] if len(sys.argv) > 1 else "13b"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SH = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6)}[name]
ll = HipLlama(SH["H"], SH["heads"], SH["I"], layers, 512, SH["eps"], pack_weights=False).init_random(seed=1)
B, S = 1, 328
cache = ll.new_cache(B, S + 64)
cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
cache.seq_len = 0
ll.forward(torch.randn((B * S, SH["H"]), device="cuda") * 0.5, B, S, cache)
decode.PERSISTENT = True
sess = decode.DecodeSession(ll, cache, use_graph=True)
sess.begin(torch.tensor([3], device="cuda"))
for _ in range(8):
    sess.step()
NWG = 256
buf = torch.zeros((NWG, layers, 5, 5), dtype=torch.int64, device="cuda")
L = lib.load()
L.vlydbg_decode_timing.argtypes = [ctypes.c_void_p]
L.vlydbg_decode_timing.restype = None
L.vlydbg_decode_timing(buf.data_ptr())
sess2 = decode.DecodeSession(ll, cache, use_graph=False)      # eager: the launch picks the hook up
sess2.pos.copy_(sess.pos); sess2.tok.copy_(sess.tok)
for _ in range(3):
    sess2._enqueue_step()
torch.cuda.synchronize()
L.vlydbg_decode_timing(None)
t = buf.cpu().numpy().astype(np.float64) * 0.01               # us
t0 = t[:, :, 0, 0].min(axis=0)                                # per layer: earliest phase-0 start
names = ["norm+qkv", "attention", "merge+o", "norm+gate|up", "down"]
print(f"{name}, {layers} layers: per-phase times in us, median over layers 1..{layers - 2} (min / median / max over the 256 workgroups)")
mid = slice(1, layers - 1)
tot = 0.0
for p in range(5):
    st, su, lp, ar, rl = (t[:, mid, p, k] for k in range(5))
    def f(x):
        return f"{np.median(x.min(0)):6.2f} {np.median(np.median(x, 0)):6.2f} {np.median(x.max(0)):6.2f}"
    # phase length as the chip sees it: latest release of this phase's barrier - latest release of the previous one
    prev = t[:, mid, p - 1, 4] if p > 0 else t[:, 0:layers - 2, 4, 4]
    span = np.median(rl.max(0) - prev.max(0))
    tot += span
    print(f"  {names[p]:14s} span {span:6.2f} | setup {f(su - st)} | loop {f(lp - su)} | drain {f(ar - lp)} | barrier wait {f(rl - ar)} | arrive skew {np.median(ar.max(0) - ar.min(0)):5.2f}")
print(f"  layer total {tot:.2f} us")
