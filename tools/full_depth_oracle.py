#!/usr/bin/env python3
"""Full-depth parity of the half-precision production path against the CPU oracle: Llama-2-7B shapes, ALL 32 decoder layers
(BASELINE.json configs[1]'s prefill: S = 328), random weights generated on the device and handed to the oracle as the
same bf16-representable values.  Three evaluations of the same problem at depths 2 / 8 / 16 / 32:
    oracle fp32 (the reference arithmetic)   |   oracle with storage rounding at the HIP pipeline's storage points
    (oracle.rounding())   |   the HIP path.
Prints one JSON object (relative L2 of the final-norm hidden state and of the logits at every depth); committed under
profiles/history/r03/.  Runs ~1 min on the GPU box, most of it the host-side oracle."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import valley_oracle as O  # noqa: E402
from valley_amd import runtime  # noqa: E402
from valley_amd.llama import HipLlama  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def run(depths=(2, 8, 16, 32), rounded=True, threads=None):
    """-> (rows, oracle seconds, threads); tests/test_depth_gpu.py runs depth 32 without the rounded evaluation."""
    H, heads, I, L, eps, S, V = 4096, 32, 11008, 32, 1e-5, 328, 512
    half = getattr(runtime, "HALF", torch.bfloat16)
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(11)
    rn = lambda *shape, std=0.02: (torch.randn(shape, generator=g, device=d) * std).to(half)  # noqa: E731
    sd = {"model.embed_tokens.weight": rn(V, H).cpu(), "lm_head.weight": rn(V, H).cpu(), "model.norm.weight": torch.ones(H)}
    for i in range(L):
        p = f"model.layers.{i}."
        for n in "qkvo":
            sd[p + f"self_attn.{n}_proj.weight"] = rn(H, H).cpu()
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = rn(I, H).cpu(), rn(I, H).cpu()
        sd[p + "mlp.down_proj.weight"] = rn(H, I).cpu()
        sd[p + "input_layernorm.weight"] = 1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(i))
        sd[p + "post_attention_layernorm.weight"] = 1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(100 + i))
    ll = HipLlama(H, heads, I, L, V, eps, device=d, pack_weights=False).load_state_dict(sd)
    emb = (torch.randn((1, S, H), generator=torch.Generator().manual_seed(3)) * 0.5)
    cfg = O.LlamaCfg(hidden=H, heads=heads, intermediate=I, layers=L, vocab=V, eps=eps)
    lm = sd["lm_head.weight"].float()
    threads = threads or int(os.environ.get("VALLEY_ORACLE_THREADS", "32"))
    torch.set_num_threads(threads)
    rows = []
    t_or = 0.0
    for k in depths:
        cache = ll.new_cache(1, S)
        x = ll.forward(emb.to(d).view(S, H).clone(), 1, S, cache, n_layers=k)
        hip_h = x.float().cpu()
        hip_l = ll.logits(x).float().cpu()
        t0 = time.perf_counter()
        with torch.no_grad():
            ref_h, _ = O.llama_forward(emb, sd, cfg, n_layers=k)
            ref_l = torch.nn.functional.linear(ref_h, lm)
            if rounded:
                with O.rounding():
                    rnd_h, _ = O.llama_forward(emb, sd, cfg, n_layers=k)
                    rnd_l = torch.nn.functional.linear(rnd_h, lm)
        t_or += time.perf_counter() - t0
        rows.append({"layers": k,
                     "hip_vs_fp32_hidden": round(rel(hip_h, ref_h[0]), 5), "hip_vs_fp32_logits": round(rel(hip_l, ref_l[0]), 5),
                     "logits_max_abs_hip_vs_fp32": round(float((hip_l - ref_l[0]).abs().max()), 4),
                     "logit_abs_max": round(float(ref_l.abs().max()), 3)})
        if rounded:
            rows[-1].update({"rounded_oracle_vs_fp32_hidden": round(rel(rnd_h[0], ref_h[0]), 5),
                             "rounded_oracle_vs_fp32_logits": round(rel(rnd_l[0], ref_l[0]), 5),
                             "hip_vs_rounded_oracle_logits": round(rel(hip_l, rnd_l[0]), 5)})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    return rows, t_or, threads, half


def main():
    rows, t_or, threads, half = run()
    print(json.dumps({"what": "Llama-2-7B shapes, 32 layers, S = 328, B = 1: HIP path vs oracle fp32 vs oracle.rounding()",
                      "storage_dtype": str(half), "oracle_seconds": round(t_or, 1), "oracle_threads": threads, "depths": rows}))


if __name__ == "__main__":
    main()
