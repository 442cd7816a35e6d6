"""Parity metrics of the half-precision production path under THIS process's VALLEY_PRECISION (bf16 or fp16): run by
tests/test_fp16_gpu.py once per storage type (the 16-bit type is a property of the loaded library, so each type needs its
own process).  Prints one JSON object on the last line of stdout."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import golden_cfg as G  # noqa: E402
from tests.test_model_gpu import build_golden_model, maxabs, rel  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    from valley_amd import lib, ops, runtime
    from valley_amd.decode import DecodeSession
    res = {"precision": runtime.PRECISION, "storage": str(runtime.HALF), "library": os.path.basename(lib.lib_path()),
           "vly_storage_dtype": lib.load().vly_storage_dtype()}
    # ---- (1) the reference's own fixtures: tower, spliced embeddings, logits (mean pooling), greedy tokens
    model = build_golden_model("mean")
    g1 = np.load(os.path.join(GOLD, "g1_tower.npz"))
    px = torch.from_numpy(G.golden_pixels(2, "g1"))
    sel = model.get_model().vision_tower.encode(px.cuda(), select_layer=-2).cpu().numpy()
    res["tower_rel"] = rel(sel[:1], g1["hs_sel_full"])
    g2 = np.load(os.path.join(GOLD, "g2_forward_mean.npz"))
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224).cuda()
    emb = model.get_model().embed_inputs(torch.from_numpy(ids), images).view(2, -1, G.GCFG["H"]).cpu().numpy()
    res["embeds_rel"] = rel(emb, g2["embeds"])
    out = model(input_ids=torch.from_numpy(ids).cuda(), images=images, attention_mask=torch.from_numpy(mask).cuda())
    v = mask.astype(bool)
    lg = out.logits.cpu().numpy()
    res["golden_logits_maxabs"], res["golden_logits_rel"] = maxabs(lg[v], g2["logits"][v]), rel(lg[v], g2["logits"][v])
    g5 = np.load(os.path.join(GOLD, "g5_decode2.npz"))
    ids5, _ = G.golden_ids("decode2")
    img1 = torch.from_numpy(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224).cuda()
    seq = model.generate(torch.from_numpy(ids5).cuda(), images=img1, max_new_tokens=8, use_graph=True)
    res["greedy_tokens_match_reference"] = seq[0, ids5.shape[1]:].tolist() == g5["tokens"][0].tolist()
    del model
    # ---- (2) 13B layer shapes, 2 layers + head, left-padded batch, vs the fp32 path on the same weights; one graph decode step
    from tests.test_depth_gpu import H, V, _pair
    ll, pl = _pair(2, seed=41)
    B, S = 2, 336
    gen = torch.Generator(device="cuda").manual_seed(7)
    e = torch.randn((B * S, H), generator=gen, device="cuda") * 0.5
    valid = torch.ones((B, S + 8), dtype=torch.uint8, device="cuda")
    valid[1, :11] = 0
    c16, c32 = ll.new_cache(B, S + 8), pl.new_cache(B, S + 8)
    c16.key_valid, c32.key_valid = valid.clone(), valid.clone()
    x16, x32 = ll.forward(e.clone(), B, S, c16), pl.forward(e.clone(), B, S, c32)
    vv = valid[:, :S].bool().reshape(-1)
    l16, l32 = ll.logits(x16)[vv], pl.logits(x32)[vv]
    res["shape13b_hidden_rel"] = float((x16.float()[vv] - x32[vv]).norm() / x32[vv].norm())
    res["shape13b_logits_rel"] = float((l16 - l32).norm() / l32.norm())
    res["shape13b_logits_maxabs"] = float((l16 - l32).abs().max())
    res["shape13b_logit_absmax"] = float(l32.abs().max())
    tok = torch.tensor([3, 7], device="cuda")
    sess = DecodeSession(ll, c16, use_graph=True)
    sess.begin(tok)
    sess.step()
    d32 = pl.logits(pl.forward(pl.embed[tok].clone(), B, 1, c32))
    res["shape13b_decode_step_rel"] = float((sess.logits[:, :V] - d32).norm() / d32.norm())
    del ll, pl
    # ---- (3) one rounding of the result: GEMM with 16-bit output vs the fp32 product of the same operands
    gen = torch.Generator(device="cuda").manual_seed(9)
    for (M, N, K, epi) in ((2688, 5120, 5120, ops.EPI_NONE), (8224, 4096, 1024, ops.EPI_QUICK_GELU), (1312, 22016, 4096, ops.EPI_SWIGLU)):
        a = torch.randn((M, K), generator=gen, device="cuda").to(runtime.HALF)
        w = (torch.randn((N, K), generator=gen, device="cuda") * 0.05).to(runtime.HALF)
        y = ops.gemm(a, w, epilogue=epi).float()
        ref = a.float() @ w.float().t()
        if epi == ops.EPI_QUICK_GELU:
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == ops.EPI_SWIGLU:
            ref = torch.nn.functional.silu(ref[:, 0::2]) * ref[:, 1::2]
        res[f"gemm_{M}x{N}x{K}_e{epi}_rel"] = float((y - ref).norm() / ref.norm())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
