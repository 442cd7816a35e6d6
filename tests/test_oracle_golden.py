"""Pin the CPU oracle (oracle/valley_oracle.py) to the fixtures captured from the reference
(tools/gen_goldens.py -> tests/golden/).  fp32 vs fp32, tolerance 2e-5 max-abs."""
import os

import numpy as np
import pytest
import torch

from oracle import valley_oracle as O
from tests import golden_cfg as G

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5


def cfgs():
    c = G.GCFG
    lcfg = O.LlamaCfg(hidden=c["H"], heads=c["heads"], intermediate=c["I"], layers=c["L"], vocab=c["vocab"], eps=c["eps"])
    vcfg = O.VisionCfg(intermediate=c["VI"], layers=c["VL"])
    tok = O.TokenIds(**G.special())
    return lcfg, vcfg, tok


def maxabs(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max())


def test_tower_hidden_states():
    g = np.load(os.path.join(GOLD, "g1_tower.npz"))
    _, vcfg, _ = cfgs()
    px = torch.from_numpy(G.golden_pixels(2, "g1"))
    hs = O.clip_hidden_states(px, G.vision_state(), vcfg)
    assert len(hs) == vcfg.layers + 1
    for i, h in enumerate(hs):
        assert maxabs(h.numpy()[:, ::8, ::4], g[f"hs{i}"]) < TOL, i
    sel = O.vit_select(px, G.vision_state(), vcfg, -2)
    assert maxabs(sel.numpy()[:1], g["hs_sel_full"]) < TOL


@pytest.mark.parametrize("method", ["mean", "max", "temporal_importance", "temporal_transformer"])
def test_forward_all_poolings(method):
    g = np.load(os.path.join(GOLD, f"g2_forward_{method}.npz"))
    lcfg, vcfg, tok = cfgs()
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    assert (ids == g["ids"]).all() and (mask == g["mask"]).all()
    w = dict(G.llama_state())
    w.update(G.extra_pool_state(method))
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    logits, _, emb = O.valley_forward(torch.from_numpy(ids), images, w, G.vision_state(), lcfg, vcfg, tok,
                                      attention_mask=torch.from_numpy(mask), method=method)
    assert maxabs(emb.numpy(), g["embeds"]) < TOL
    valid = mask.astype(bool)                 # padded query rows are don't-care
    ref = g["logits"]
    got = logits.numpy() if method == "mean" else logits.numpy()[:, ::4]
    v = valid if method == "mean" else valid[:, ::4]
    assert maxabs(got[v], ref[v]) < 5e-5


@pytest.mark.parametrize("case", ["mixed", "two_images", "frame_mismatch"])
def test_splice_cases(case):
    g = np.load(os.path.join(GOLD, f"g3_{case}.npz"))
    lcfg, vcfg, tok = cfgs()
    T = G.GCFG["T"]
    ids, mask = G.golden_ids(case)
    img1 = torch.from_numpy(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224)
    logits, _, emb = O.valley_forward(torch.from_numpy(ids), img1, G.llama_state(), G.vision_state(), lcfg, vcfg, tok,
                                      attention_mask=torch.from_numpy(mask))
    assert maxabs(emb.numpy(), g["embeds"]) < TOL
    v = mask.astype(bool)[:, ::4]
    assert maxabs(logits.numpy()[:, ::4][v], g["logits"][v]) < 5e-5


def test_splice_errors():
    g = np.load(os.path.join(GOLD, "g3_errors.npz"))
    lcfg, vcfg, tok = cfgs()
    T = G.GCFG["T"]
    img1 = torch.from_numpy(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224)
    for case in ("cut", "unbalanced"):
        ids, mask = G.golden_ids(case)
        with pytest.raises(ValueError) as e:
            O.valley_forward(torch.from_numpy(ids), img1, G.llama_state(), G.vision_state(), lcfg, vcfg, tok)
        assert str(g[case]) == f"ValueError: {e.value}"


def test_list_of_clips():
    g = np.load(os.path.join(GOLD, "g3_list.npz"))
    lcfg, vcfg, tok = cfgs()
    ids, mask = G.golden_ids("list")
    clips = [torch.from_numpy(G.golden_pixels(2, "list0")), torch.from_numpy(G.golden_pixels(3, "list1"))]
    logits, _, emb = O.valley_forward(torch.from_numpy(ids), clips, G.llama_state(), G.vision_state(), lcfg, vcfg, tok,
                                      attention_mask=torch.from_numpy(mask))
    assert maxabs(emb.numpy(), g["embeds"]) < TOL
    v = mask.astype(bool)[:, ::4]
    assert maxabs(logits.numpy()[:, ::4][v], g["logits"][v]) < 5e-5


def test_greedy_decode_loop():
    g = np.load(os.path.join(GOLD, "g5_decode.npz"))
    lcfg, vcfg, tok = cfgs()
    T = G.GCFG["T"]
    ids, _ = G.golden_ids("decode")
    img1 = torch.from_numpy(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224)
    toks, lasts = O.greedy_decode(torch.from_numpy(ids), img1, G.llama_state(), G.vision_state(), lcfg, vcfg, tok, 4)
    assert (toks.numpy() == g["tokens"]).all()
    assert maxabs(lasts.numpy(), g["last_logits"]) < 5e-5


def test_attention_probabilities_vs_reference_fixture():
    """``output_attentions=True`` (valley_model.py:281, 324-330 -> HF LlamaModel's all_self_attns, eager attention): the oracle's
    per-layer softmax probabilities against the reference's own tuple (g11, tools/gen_goldens_r4.py) — B = 2, one row
    left-padded; rows of padded queries are don't-care (HF leaves a uniform row there)."""
    g = np.load(os.path.join(GOLD, "g11_attentions.npz"))
    lcfg, vcfg, tok = cfgs()
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    at = []
    with torch.no_grad():
        logits, _, _ = O.valley_forward(torch.from_numpy(ids), images, dict(G.llama_state()), G.vision_state(), lcfg, vcfg, tok,
                                        attention_mask=torch.from_numpy(mask), attn_out=at)
    assert len(at) == int(g["n"]) == lcfg.layers
    v = mask.astype(bool)[:, ::4]                            # valid QUERY rows of the sub-sampled fixture
    S = ids.shape[1]
    for i, a in enumerate(at):
        assert tuple(a.shape) == (2, lcfg.heads, S, S)
        got, ref = a.numpy()[:, :, ::4], g[f"attn{i}"]
        for b in range(2):
            assert maxabs(got[b][:, v[b]], ref[b][:, v[b]]) < TOL, (i, b)
            assert np.all(got[b][:, v[b]][..., ~mask.astype(bool)[b]] == 0)      # padded keys get exactly zero
    assert maxabs(logits.numpy()[:, ::4][v], g["logits"][v]) < 5e-5
