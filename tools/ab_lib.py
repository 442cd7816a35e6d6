#!/usr/bin/env python3
"""In-process A/B of GEMM kernels from several builds of libvalley_hip.so (cdna_hip_programming: any claim < 5 %
needs an interleaved A/B inside one probe).

  build (CPU):  python tools/ab_lib.py build NAME -DVLY_FRAG_ORDER=1 ...   -> valley_amd/lib/variants/libvalley_hip_NAME.so
                python tools/ab_lib.py build NAME --src attention.hip -DVLY_ATTN_ORDER=0
  run   (GPU):  python tools/ab_lib.py run base,NAME[,NAME2] [shape ...]
                python tools/ab_lib.py run-attn base,NAME [B,S,heads ...]

A shape is M,N,K,epi,tile[|tile...] ("1312,22016,4096,2,8|105"); every (library, tile) pair is an arm; defaults = the
hot-path shapes with their shipped tiles.  Every
library is called through the C ABI directly (ctypes), round-robin per repetition, on operands that rotate through
four weight copies (every call reads weights that left the Infinity Cache), arms in a fresh random order per repetition;
every library's result on the same operands is checked against the first library's.
"""
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARDIR = os.path.join(ROOT, "valley_amd", "lib", "variants")

DEFAULT_SHAPES = ["1312,22016,4096,2,8", "1312,12288,4096,0,86", "8224,4096,1024,1,8", "8224,3072,1024,0,9",
                  "2688,27648,5120,2,9", "1312,4096,11008,0,7", "1312,12288,4096,0,76"]


def build(name, flags):
    from valley_amd import build as b
    os.makedirs(VARDIR, exist_ok=True)
    b.build(verbose=False)
    src = "gemm_bf16.hip"
    if flags and flags[0] == "--src":                       # the one source file the flags apply to
        src, flags = flags[1], flags[2:]
    obj = os.path.join(VARDIR, f"{src[:-4]}_{name}.o")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c",
                           os.path.join(b.CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES]
    out = os.path.join(VARDIR, f"libvalley_hip_{name}.so")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


def run(names, shapes, reps=40):
    import torch
    from valley_amd import build as b
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = []
    for n in names:
        L = ctypes.CDLL(b.LIB if n == "base" else os.path.join(VARDIR, f"libvalley_hip_{n}.so"))
        L.vly_gemm_bf16.restype = I
        L.vly_gemm_bf16.argtypes = [P] * 5 + [I] * 10 + [P]
        libs.append(L)
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    import random
    rng = random.Random(0)
    for sh in shapes:
        M, N, K, epi, tiles = sh.split(",")
        M, N, K, epi = int(M), int(N), int(K), int(epi)
        arms = [(li, int(t)) for li in range(len(libs)) for t in tiles.split("|")]      # an arm = (library, tile hint)
        a = torch.randn((M, K), device=d).to(torch.bfloat16)
        ws = [(torch.randn((N, K), device=d) * 0.05).to(torch.bfloat16) for _ in range(4)]
        # libraries named packed*: built with -DVLY_W_PACKED=1, weights as [K/64][N/64][64][64] blocks
        wp = [w.view(N // 64, 64, K // 64, 64).permute(2, 0, 1, 3).contiguous() for w in ws] \
            if any(n.startswith("packed") for n in names) and N % 64 == 0 else None
        No = N // 2 if epi == 2 else N
        outs = [torch.empty((M, No), device=d, dtype=torch.bfloat16) for _ in arms]
        times = [[] for _ in arms]

        def call(ai, wi):
            li, t = arms[ai]
            w = wp[wi] if names[li].startswith("packed") else ws[wi]
            rc = libs[li].vly_gemm_bf16(a.data_ptr(), w.data_ptr(), None, None, outs[ai].data_ptr(), M, N, K, K, K, No, 0, epi,
                                        0, t, st)
            assert rc == 0, (names[li], t, rc)

        for ai in range(len(arms)):                                        # same operands: same result
            call(ai, 0)
        torch.cuda.synchronize()
        for ai in range(1, len(arms)):
            err = float((outs[ai].float() - outs[0].float()).norm() / outs[0].float().norm())
            if os.environ.get("AB_NOCHECK") == "1":                        # timing experiments with deliberately broken variants
                continue
            assert err < 2e-3, (names[arms[ai][0]], arms[ai][1], sh, err)
            if os.environ.get("AB_BITEXACT") == "1":                       # same summation order: the same bits
                assert torch.equal(outs[ai], outs[0]), (names[arms[ai][0]], arms[ai][1], sh, "not bit-identical")
        ncall = 0
        for r in range(reps + 3):
            order = list(range(len(arms)))
            rng.shuffle(order)                                             # no arm keeps the same predecessor
            for ai in order:
                wi = ncall % 4                                             # last touched >= 4 calls ago: out of the MALL
                ncall += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                call(ai, wi)
                e1.record()
                torch.cuda.synchronize()
                if r >= 3:
                    times[ai].append(e0.elapsed_time(e1) * 1e3)
        fl = 2.0 * M * N * K
        row = {"shape": f"{M}x{N}x{K}/e{epi}"}
        base = statistics.median(times[0])
        for (li, t), tm in zip(arms, times):
            med = statistics.median(tm)
            row[f"{names[li]}:{t}"] = f"{med:.1f}us {fl / med / 1e6:.0f}TF {100 * (base / med - 1):+.1f}%"
        print(json.dumps(row), flush=True)


def run_attn(names, shapes, reps=60):
    """Prefill attention (vly_llama_attention) A/B: shape = B,S,heads."""
    import random
    import torch
    from valley_amd import build as b
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = []
    for n in names:
        L = ctypes.CDLL(b.LIB if n == "base" else os.path.join(VARDIR, f"libvalley_hip_{n}.so"))
        L.vly_llama_attention.restype = I
        L.vly_llama_attention.argtypes = [P, P, P, P, I, P, I, I, I, I, P, I, P]
        libs.append(L)
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = random.Random(0)
    for sh in shapes:
        B, S, heads = (int(x) for x in sh.split(","))
        qkv = torch.randn((B * S, 3 * heads * 128), device=d).to(torch.bfloat16)
        kc = torch.randn((B, heads, S, 128), device=d).to(torch.bfloat16)
        vc = torch.randn((B, heads, S, 128), device=d).to(torch.bfloat16)
        outs = [torch.empty((B * S, heads * 128), device=d, dtype=torch.bfloat16) for _ in libs]
        times = [[] for _ in libs]
        for r in range(reps + 3):
            order = list(range(len(libs)))
            rng.shuffle(order)
            for li in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = libs[li].vly_llama_attention(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), None, 0, outs[li].data_ptr(), B, S,
                                                  heads, 0, None, S, st)
                e1.record()
                assert rc == 0
                torch.cuda.synchronize()
                if r >= 3:
                    times[li].append(e0.elapsed_time(e1) * 1e3)
        for li in range(1, len(libs)):
            assert torch.equal(outs[li], outs[0]), names[li]
        print(json.dumps({"attn": sh, **{n: round(statistics.median(t), 1) for n, t in zip(names, times)}, "unit": "us"}), flush=True)


def run_vit_attn(names, frames, reps=int(os.environ.get("AB_REPS", "40"))):
    """ViT attention (vly_vit_attention) A/B: one arm per library, F frames per launch."""
    import random
    import torch
    from valley_amd import build as b
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = []
    for n in names:
        L = ctypes.CDLL(b.LIB if n == "base" else os.path.join(VARDIR, f"libvalley_hip_{n}.so"))
        L.vly_vit_attention.restype = I
        L.vly_vit_attention.argtypes = [P, P, I, P]
        libs.append(L)
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = random.Random(0)
    for F in frames:
        qkvs = [torch.randn((F * 257, 3072), device=d).to(torch.bfloat16) for _ in range(3)]     # rotate: cold-ish inputs
        outs = [torch.empty((F * 257, 1024), device=d, dtype=torch.bfloat16) for _ in libs]
        times = [[] for _ in libs]
        n = 0
        for r in range(reps + 3):
            order = list(range(len(libs)))
            rng.shuffle(order)
            for li in order:
                q = qkvs[n % 3]
                n += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = libs[li].vly_vit_attention(q.data_ptr(), outs[li].data_ptr(), F, st)
                e1.record()
                assert rc == 0
                torch.cuda.synchronize()
                if r >= 3:
                    times[li].append(e0.elapsed_time(e1) * 1e3)
        for li in range(len(libs)):
            libs[li].vly_vit_attention(qkvs[0].data_ptr(), outs[li].data_ptr(), F, st)
        torch.cuda.synchronize()
        errs = [float((outs[li].float() - outs[0].float()).norm() / outs[0].float().norm()) for li in range(len(libs))]
        print(json.dumps({"vit_attn_frames": F, **{n_: round(statistics.median(t), 1) for n_, t in zip(names, times)}, "unit": "us",
                          "rel_vs_first": [round(e, 5) for e in errs]}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "run-vit-attn":
        run_vit_attn(sys.argv[2].split(","), [int(x) for x in sys.argv[3:]] or [32, 128, 256])
    elif sys.argv[1] == "run-attn":
        run_attn(sys.argv[2].split(","), sys.argv[3:] or ["4,328,32", "8,336,40", "8,352,40"])
    else:
        run(sys.argv[2].split(","), sys.argv[3:] or DEFAULT_SHAPES)
