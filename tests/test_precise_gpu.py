"""BASELINE.json: "logits within 1e-3 of reference".  Three statements, each tested here (SURVEY.md §7 "hard parts", §8c):

(i)   the bf16 production path against the fp32 reference fixtures: measured and asserted in tests/test_model_gpu.py
      (max-abs 0.030 on |logit| < 4 — the cost of bf16 storage, NOT of the implementation, see (ii));
(ii)  the bf16 production path against the SAME-DTYPE oracle (oracle.rounding(): the fp32 restatement rounded to bf16 at
      exactly the points where the HIP pipeline stores bf16).  Measured: that oracle is itself 3.1e-2 (max-abs) from the
      fp32 reference, the HIP path 3.05e-2, and the two are 2.6e-2 from each other — two bf16-storage evaluations of a
      2-layer model whose summation orders differ decorrelate (a 1-ulp flip of a bf16 rounding is a 4e-3 relative
      perturbation that the next layers carry).  So the testable statement is: the HIP path's error against the fp32
      reference does not exceed the error of the oracle's bf16-storage evaluation (x1.15) — asserted below;
(iii) VALLEY_PRECISION=fp32 (valley_amd/precise.py: fp32 tensors end to end, exact f32 MFMA) against the fp32 reference
      fixtures and the fp32 oracle: **max-abs logit error < 1e-3** asserted on the golden model (all of its cases) and
      on 2 decoder layers at the Vicuna-13B layer shapes.

Plus unit parity of every fp32 kernel (vly_*_f32) against a plain PyTorch fp32 statement of the op."""
import os

import numpy as np
import pytest
import torch

from tests import golden_cfg as G

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL_FP32 = 1e-3          # the north star's bound, fp32 mode
D = "cuda:0"


def maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# ---- fp32 kernels -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (64, 64, 16), (1, 512, 592), (771, 1028, 1024), (130, 36, 48)])
def test_gemm_f32(M, N, K):
    from valley_amd import ops, ops_f32 as F
    a, w, bias, res = rnd((M, K), 1), rnd((N, K), 2, 0.05), rnd((N,), 3, 0.5), rnd((M, N), 4)
    base = (a.double() @ w.double().t() + bias.double())
    out = F.gemm(a.to(D), w.to(D), bias.to(D))
    assert maxabs(out.cpu(), base) < 2e-5 * max(1.0, float(base.abs().max()))
    h = res.to(D).clone()
    F.gemm(a.to(D), w.to(D), bias.to(D), residual=h, out=h)                        # in-place residual update
    assert maxabs(h.cpu(), base + res.double()) < 3e-5 * max(1.0, float(base.abs().max()))
    out = F.gemm(a.to(D), w.to(D), bias.to(D), epilogue=ops.EPI_QUICK_GELU)
    assert maxabs(out.cpu(), base * torch.sigmoid(1.702 * base)) < 3e-5 * max(1.0, float(base.abs().max()))
    out = F.gemm(a.to(D), w.to(D), bias.to(D), epilogue=ops.EPI_RELU)
    assert maxabs(out.cpu(), base.clamp_min(0)) < 3e-5 * max(1.0, float(base.abs().max()))
    if N % 8 == 0:
        nb = a.double() @ w.double().t()
        out = F.gemm(a.to(D), w.to(D), epilogue=ops.EPI_SWIGLU)
        assert out.shape == (M, N // 2)
        assert maxabs(out.cpu(), torch.nn.functional.silu(nb[:, 0::2]) * nb[:, 1::2]) < 3e-5 * max(1.0, float(nb.abs().max()) ** 2)
    # transpose detector
    w1 = torch.zeros((N, K))
    w1[3, K - 1] = 1.0
    out = F.gemm(a.to(D), w1.to(D))
    assert maxabs(out.cpu()[:, 3], a[:, K - 1]) == 0.0 and float(out.cpu()[:, :3].abs().max()) == 0.0


def test_vit_attention_f32():
    from valley_amd import ops_f32 as F
    Fn = 3
    qkv = rnd((Fn * 257, 3072), 5)
    qkv[5, :64] *= 6.0
    out = F.vit_attention(qkv.to(D), Fn).cpu()
    x = qkv.double().view(Fn, 257, 3, 16, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(Fn * 257, 1024)
    assert maxabs(out, ref) < 2e-5


@pytest.mark.parametrize("B,S,heads,hd,causal,nz,past", [(2, 75, 2, 64, 0, 0, 0), (2, 75, 2, 64, 1, 5, 0), (3, 257, 16, 64, 0, 0, 0), (2, 75, 2, 128, 0, 0, 0),
                                                       (2, 336, 4, 128, 1, 11, 0), (1, 16, 1, 128, 0, 3, 0), (2, 64, 2, 128, 1, 0, 0), (2, 40, 2, 128, 1, 7, 100)])
def test_attention_f32_on_the_f32_mfma(B, S, heads, hd, causal, nz, past):
    """vly_attention_f32 with 16 queries or more = the tiled kernel on v_mfma_f32_16x16x4_f32 (round 6): head dims 64 / 128, causal or not,
    invalid leading keys (left padding), a past, query counts that are not multiples of the 64-query workgroup — against fp64 softmax
    attention, and bit-identical from run to run (the wave-private P strip is fenced)."""
    from valley_amd import lib as _lib
    L = _lib.load()
    ctx = 512
    n_kv = past + S
    q, k, v = rnd((B, S, heads, hd), 1).to(D), rnd((B, heads, ctx, hd), 2).to(D), rnd((B, heads, ctx, hd), 3).to(D)
    valid = torch.ones((B, ctx), dtype=torch.uint8, device=D)
    valid[0, :nz] = 0
    outs = []
    for _ in range(2):
        out = torch.full((B, S, heads, hd), float("nan"), device=D)
        rc = L.vly_attention_f32(q.data_ptr(), S * heads * hd, heads * hd, k.data_ptr(), v.data_ptr(), heads * ctx * hd, ctx * hd, hd,
                                 valid.data_ptr() if nz else None, ctx, out.data_ptr(), S * heads * hd, heads * hd, B, heads, S, n_kv, hd,
                                 causal, past, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    s = (q.cpu().transpose(1, 2).double() @ k.cpu()[:, :, :n_kv].double().transpose(-1, -2)) * hd ** -0.5
    allowed = torch.ones((B, 1, S, n_kv), dtype=torch.bool)
    if causal:
        allowed = allowed & (torch.arange(n_kv)[None, :] <= torch.arange(S)[:, None] + past)[None, None]
    allowed = allowed & valid.cpu()[:, None, None, :n_kv].bool()
    p = torch.softmax(torch.where(allowed, s, torch.tensor(-1e30, dtype=torch.float64)), -1)
    ref = (p @ v.cpu()[:, :, :n_kv].double()).transpose(1, 2)
    seen = allowed.any(-1).transpose(1, 2)[..., None]                    # fully masked query rows: zeros by contract
    assert float((outs[0].double() * (~seen)).abs().max()) == 0.0
    e = float(((outs[0].double() - ref).abs() * seen).max())
    print(f"attention_f32 (MFMA) B{B} S{S} heads{heads} hd{hd} causal{causal} invalid{nz} past{past}: max err {e:.1e}")
    assert e < 5e-6


@pytest.mark.parametrize("B,S,past,heads,pad", [(2, 75, 0, 2, 9), (1, 130, 0, 3, 0), (2, 1, 130, 2, 5), (1, 40, 100, 2, 0)])
def test_rope_and_llama_attention_f32(B, S, past, heads, pad):
    from valley_amd import ops_f32 as F
    H, ctx = heads * 128, 256
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    ang = torch.arange(ctx, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    kc, vc = torch.zeros((B, heads, ctx, 128)), torch.zeros((B, heads, ctx, 128))
    if past:
        kc[:, :, :past], vc[:, :, :past] = rnd((B, heads, past, 128), 6), rnd((B, heads, past, 128), 7)
    qkv = rnd((B * S, 3 * H), 8)
    qd, kd, vd = qkv.to(D).clone(), kc.to(D), vc.to(D)
    F.rope_kv(qd, kd, vd, cos.to(D), sin.to(D), B, S, heads, past)
    x = qkv.double().view(B, S, 3, heads, 128)
    pos = torch.arange(S) + past
    c = torch.cat([cos[pos], cos[pos]], -1).double()[None, :, None]
    sn = torch.cat([sin[pos], sin[pos]], -1).double()[None, :, None]
    rot = lambda t: torch.cat([-t[..., 64:], t[..., :64]], -1)  # noqa: E731
    qr, kr = x[:, :, 0] * c + rot(x[:, :, 0]) * sn, x[:, :, 1] * c + rot(x[:, :, 1]) * sn
    assert maxabs(qd.cpu().view(B, S, 3, heads, 128)[:, :, 0], qr) < 2e-6
    assert maxabs(kd.cpu()[:, :, past:past + S], kr.transpose(1, 2)) < 2e-6
    assert maxabs(vd.cpu()[:, :, past:past + S], x[:, :, 2].transpose(1, 2)) == 0.0
    kv = past + S
    valid = torch.ones((B, ctx), dtype=torch.uint8)
    if pad:
        valid[0, :pad] = 0
    out = F.llama_attention(qd, kd, vd, valid.to(D) if pad else None, B, S, heads, past).cpu()
    K, V = kd.cpu().double()[:, :, :kv], vd.cpu().double()[:, :, :kv]
    s = qr.transpose(1, 2) @ K.transpose(-1, -2) * 128 ** -0.5
    i, j = torch.arange(S)[:, None] + past, torch.arange(kv)[None]
    allowed = (j <= i)[None, None].expand(B, 1, S, kv) & valid[:, None, None, :kv].bool()
    ref = (torch.softmax(torch.where(allowed, s, torch.tensor(-1e30, dtype=torch.float64)), -1) @ V).transpose(1, 2).reshape(B * S, H)
    rows = torch.ones(B * S, dtype=torch.bool)
    if pad and past == 0:
        rows[:pad] = False                       # fully masked query rows: zeros by contract, don't-care in the reference
        assert float(out[:pad].abs().max()) == 0.0
    assert maxabs(out[rows], ref[rows]) < 2e-5


def test_norm_patchify_pool_embed_f32():
    from valley_amd import ops, ops_f32 as F
    x, g, b = rnd((37, 5120), 9, 2.0) + 0.3, rnd((5120,), 10, 0.1) + 1.0, rnd((5120,), 11, 0.1)
    assert maxabs(F.norm(x.to(D), g.to(D), b.to(D), 1e-5).cpu(), torch.nn.functional.layer_norm(x.double(), (5120,), g.double(), b.double(), 1e-5)) < 2e-5
    ref = g.double() * (x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6))
    assert maxabs(F.norm(x.to(D), g.to(D), None, 1e-6).cpu(), ref) < 2e-5
    img = rnd((2, 3, 224, 224), 12)
    cols = F.patchify(img.to(D)).cpu()
    ref_cols = torch.nn.functional.unfold(img, kernel_size=14, stride=14).transpose(1, 2).reshape(512, 588)
    assert maxabs(cols[:, :588], ref_cols) == 0.0 and float(cols[:, 588:].abs().max()) == 0.0 and cols.shape[1] == 592
    Bc, T, W = 2, 5, 256
    f = rnd((Bc, T, 257, W), 13)
    sc = rnd((Bc * T,), 14)
    for mode, pooled in ((ops.POOL_MEAN, f[:, :, 1:].mean(1)), (ops.POOL_MAX, f[:, :, 1:].max(1)[0]),
                         (ops.POOL_IMPORTANCE, (torch.softmax(sc.view(Bc, T), 1)[:, :, None, None] * f[:, :, 1:]).sum(1))):
        out = F.pool_tokens(f.to(D).view(-1, W), Bc, T, mode, sc.to(D) if mode == ops.POOL_IMPORTANCE else None).cpu()
        assert maxabs(out, torch.cat([pooled, f[:, :, 0]], 1)) < 2e-6
    emb, vis = rnd((50, 256), 15), rnd((7, 256), 16)
    rmap = torch.tensor([0, 49, -1, -7, 3, -2, 10], dtype=torch.int32)
    out = F.embed_splice(rmap.to(D), emb.to(D), vis.to(D)).cpu()
    assert maxabs(out, torch.stack([emb[v] if v >= 0 else vis[-v - 1] for v in rmap.tolist()])) == 0.0


# ---- the whole path in fp32 mode --------------------------------------------------------------------------------------
def build_model(method="mean", precision="fp32"):
    from valley_amd import valley_model as vm
    c = G.GCFG
    cfg = vm.ValleyConfig(vocab_size=c["vocab"], hidden_size=c["H"], intermediate_size=c["I"], num_hidden_layers=c["L"],
                          num_attention_heads=c["heads"], num_key_value_heads=c["heads"], rms_norm_eps=c["eps"], max_position_embeddings=2048)
    cfg.use_mm_proj, cfg.mm_hidden_size, cfg.mm_vision_select_layer, cfg.valley_precision = True, 1024, -2, precision
    model = vm.ValleyLlamaForCausalLM(cfg)
    sd = dict(G.llama_state())
    sd.update(G.extra_pool_state(method))
    model.load_state_dict(sd)
    tower = vm.build_vision_tower(dict(intermediate_size=c["VI"], num_hidden_layers=c["VL"]), state_dict=G.vision_state(), precision=precision)
    for k, v in G.special().items():
        setattr(tower.config, k, v)
    model.get_model().vision_tower = tower
    model.get_model().patch_pooling_method = method
    return model


def test_fp32_mode_tower_vs_reference_fixture():
    g = np.load(os.path.join(GOLD, "g1_tower.npz"))
    tower = build_model().get_model().vision_tower
    assert tower.precision == "fp32"
    px = torch.from_numpy(G.golden_pixels(2, "g1"))
    sel = tower.encode(px.cuda(), select_layer=-2).cpu().numpy()
    print("fp32 tower max-abs", maxabs(sel[:1], g["hs_sel_full"]))
    assert maxabs(sel[:1], g["hs_sel_full"]) < 2e-4                      # |x| up to ~15
    for i, h in enumerate(tower(px.cuda(), output_hidden_states=True).hidden_states):
        assert maxabs(h.cpu().numpy()[:, ::8, ::4], g[f"hs{i}"]) < 2e-4, i


@pytest.mark.parametrize("method", ["mean", "max", "temporal_importance", "temporal_transformer"])
def test_fp32_mode_logits_within_1e3_of_reference(method):
    """The north-star bound on the reference-captured fixtures: B = 2, left-padded, mixed text / visual tokens."""
    g = np.load(os.path.join(GOLD, f"g2_forward_{method}.npz"))
    model = build_model(method)
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224).cuda()
    emb = model.get_model().embed_inputs(torch.from_numpy(ids), images).view(2, -1, G.GCFG["H"]).cpu().numpy()
    out = model(input_ids=torch.from_numpy(ids).cuda(), images=images, attention_mask=torch.from_numpy(mask).cuda())
    logits = out.logits.cpu().numpy()
    v = mask.astype(bool)
    got, vv = (logits, v) if method == "mean" else (logits[:, ::4], v[:, ::4])
    e_emb, e_log = maxabs(emb, g["embeds"]), maxabs(got[vv], g["logits"][vv])
    print(f"fp32 mode / {method}: spliced-embedding max-abs {e_emb:.2e}, logits max-abs {e_log:.2e} (bound {TOL_FP32})")
    assert e_emb < 2e-4 and e_log < TOL_FP32


def test_fp32_mode_splice_cases_and_decode_vs_reference():
    model = build_model()
    T = G.GCFG["T"]
    img1 = torch.from_numpy(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224).cuda()
    for case in ("mixed", "two_images", "frame_mismatch"):
        g = np.load(os.path.join(GOLD, f"g3_{case}.npz"))
        ids, mask = G.golden_ids(case)
        out = model(input_ids=torch.from_numpy(ids).cuda(), images=img1, attention_mask=torch.from_numpy(mask).cuda())
        v = mask.astype(bool)[:, ::4]
        assert maxabs(out.logits.cpu().numpy()[:, ::4][v], g["logits"][v]) < TOL_FP32, case
    # prefill + 8 KV-cache decode steps (the generic forward on the fp32 cache) vs the reference loop
    g = np.load(os.path.join(GOLD, "g5_decode2.npz"))
    ids, _ = G.golden_ids("decode2")
    out = model(input_ids=torch.from_numpy(ids).cuda(), images=img1, use_cache=True)
    past, logits, worst = out.past_key_values, out.logits, 0.0
    for step in range(8):
        last = logits[:, -1].cpu().numpy()
        worst = max(worst, maxabs(last, g["last_logits"][:, step]))
        assert int(last.argmax()) == int(g["tokens"][0, step])
        assert past[0][0].shape[-2] == ids.shape[1] + step and past[0][0].dtype == torch.float32
        tok = torch.from_numpy(g["tokens"][:, step]).cuda()
        o = model(input_ids=tok[:, None], use_cache=True, past_key_values=past,
                  attention_mask=torch.ones(1, past.get_seq_length() + 1, dtype=torch.long).cuda())
        logits, past = o.logits, o.past_key_values
    print("fp32 mode decode: worst last-position logit error over 8 steps", worst)
    assert worst < TOL_FP32
    seq = model.generate(torch.from_numpy(ids).cuda(), images=img1, max_new_tokens=8)
    assert seq[0, ids.shape[1]:].tolist() == g["tokens"][0].tolist()


def test_fp32_mode_13b_layer_shapes_within_1e3_of_oracle():
    """2 decoder layers + final norm + lm_head at the Vicuna-13B layer shapes (H 5120, 40 heads, I 13824, eps 1e-6),
    B = 2 x S = 336 with left padding, fp32 mode vs the fp32 oracle: the 1e-3 bound at production width."""
    from oracle import valley_oracle as O
    from valley_amd import weights as W
    from valley_amd.precise import PreciseLlama
    H, heads, I, V, B, S = 5120, 40, 13824, 512, 2, 336
    sd = W.valley_llama_weights(5, V, H, I, 2)
    ll = PreciseLlama(H, heads, I, 2, V, 1e-6).load_state_dict(sd)
    emb = W.det_normal(9, "emb.13b", (B, S, H), 0.5)
    mask = np.ones((B, S), np.int64)
    mask[1, :11] = 0
    cache = ll.new_cache(B, S)
    cache.key_valid = torch.from_numpy(mask).to(torch.uint8).cuda()
    x = ll.forward(torch.from_numpy(emb).cuda().view(B * S, H).clone(), B, S, cache)
    got = ll.logits(x).view(B, S, -1).cpu().numpy()
    with torch.no_grad():
        ref_h, _ = O.llama_forward(torch.from_numpy(emb), sd, O.LlamaCfg(hidden=H, heads=heads, intermediate=I, layers=2, vocab=V, eps=1e-6),
                                   torch.from_numpy(mask))
        ref = torch.nn.functional.linear(ref_h, torch.from_numpy(sd["lm_head.weight"])).numpy()
    v = mask.astype(bool)
    e = maxabs(got[v], ref[v])
    print(f"fp32 mode, 13B layer shapes: logits max-abs {e:.2e} (|logit| max {float(np.abs(ref[v]).max()):.2f}), rel-L2 {rel(got[v], ref[v]):.1e}")
    assert e < TOL_FP32


# ---- (i-b) the same bound on the split-operand engine (VALLEY_F32_GEMM=x3, round 6) -----------------------------------------------
@pytest.fixture
def x3_mode():
    """The fp32 engines with every GEMM as three 16-bit partial products (hi.hi + hi.lo + lo.hi) on vly_gemm_bf16 (ops_f32.GEMM_MODE)."""
    from valley_amd import ops_f32, runtime
    if runtime.HALF != torch.bfloat16:
        pytest.skip("the split-operand GEMMs are built on the bf16 library")
    ops_f32.set_gemm_mode("x3")
    yield
    ops_f32.set_gemm_mode("exact")


@pytest.mark.parametrize("M,N,K,epi", [(300, 264, 128, 0), (771, 1028, 1024, 0), (2056, 4096, 1024, 1), (672, 2048, 1024, 2), (512, 1024, 592, 0),
                                       (4, 512, 256, 0)])
def test_x3_gemm_vs_fp64(M, N, K, epi, x3_mode):
    """a . w^T as hi.hi + hi.lo + lo.hi: relative error ~2^-16 per term, against the fp64 product of the SAME fp32 operands; the
    activation of a producing GEMM is applied by the consumer's operand split (X3Act) and checked through a second GEMM."""
    from valley_amd import ops_f32 as F
    a = rnd((M, K), 1).cuda()
    w = rnd((N, K), 2, K ** -0.5).cuda()
    b = rnd((N,), 3).cuda()
    r = rnd((M, N), 4).cuda()
    ref = a.double() @ w.double().t() + b.double()
    if epi == 0:
        got = F.gemm(a, w, b, residual=r)
        ref = ref + r.double()
        e = float((got.double() - ref).abs().max() / ref.abs().max())
        print(f"x3 gemm {M}x{N}x{K}: max error / max |c| = {e:.2e}")
        assert e < 4e-5
        return
    act = F.gemm(a, w, b, epilogue=epi)
    assert isinstance(act, F.X3Act)
    if epi == 1:
        mid = ref / (1 + torch.exp(-1.702 * ref))
    else:
        mid = ref[:, 0::2] / (1 + torch.exp(-ref[:, 0::2])) * ref[:, 1::2]
    w2 = rnd((512, mid.shape[1]), 5, mid.shape[1] ** -0.5).cuda()
    got = F.gemm(act, w2)
    ref2 = mid @ w2.double().t()
    e = float((got.double() - ref2).abs().max() / ref2.abs().max())
    print(f"x3 gemm {M}x{N}x{K}, epilogue {epi} applied by the next GEMM's split: max error / max |c| = {e:.2e}")
    assert e < 6e-5


def test_x3_mode_tower_vs_reference_fixture(x3_mode):
    test_fp32_mode_tower_vs_reference_fixture()


@pytest.mark.parametrize("method", ["mean", "max", "temporal_importance", "temporal_transformer"])
def test_x3_mode_logits_within_1e3_of_reference(method, x3_mode):
    """BASELINE.json's "logits within 1e-3 of reference" on the engine that runs at a third (not a sixteenth) of the production rate."""
    test_fp32_mode_logits_within_1e3_of_reference(method)


def test_x3_mode_splice_cases_and_decode_vs_reference(x3_mode):
    test_fp32_mode_splice_cases_and_decode_vs_reference()


def test_x3_mode_13b_layer_shapes_within_1e3_of_oracle(x3_mode):
    test_fp32_mode_13b_layer_shapes_within_1e3_of_oracle()


# ---- (ii) the bf16 path against the same-dtype oracle --------------------------------------------------------------------
@pytest.mark.parametrize("method", ["mean", "max"])
def test_bf16_path_vs_same_dtype_oracle(method):
    """The production bf16 path vs oracle.rounding() (bf16 rounding at the pipeline's storage points): what remains is fp32
    summation order and the online-softmax form.  Also prints the distance of both to the fp32 reference fixture."""
    from oracle import valley_oracle as O
    from valley_amd import runtime
    if runtime.HALF != torch.bfloat16:
        pytest.skip("oracle.rounding() emulates bf16 storage; this process is bound to the fp16 library")
    g = np.load(os.path.join(GOLD, f"g2_forward_{method}.npz"))
    model = build_model(method, precision="bf16")
    c, T = G.GCFG, G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    px = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    out = model(input_ids=torch.from_numpy(ids).cuda(), images=px.cuda(), attention_mask=torch.from_numpy(mask).cuda())
    got = out.logits.cpu().numpy()
    lcfg = O.LlamaCfg(hidden=c["H"], heads=c["heads"], intermediate=c["I"], layers=c["L"], vocab=c["vocab"], eps=c["eps"])
    vcfg = O.VisionCfg(intermediate=c["VI"], layers=c["VL"])
    with torch.no_grad(), O.rounding():
        same, _, _ = O.valley_forward(torch.from_numpy(ids), px, G.llama_state(), G.vision_state(), lcfg, vcfg, O.TokenIds(**G.special()),
                                      torch.from_numpy(mask), method=method)
    same = same.numpy()
    v = mask.astype(bool)
    ref = g["logits"] if method == "mean" else None
    e_same = maxabs(got[v], same[v])
    msg = f"bf16 path / {method}: vs same-dtype oracle max-abs {e_same:.2e} rel-L2 {rel(got[v], same[v]):.1e}"
    if ref is not None:
        msg += f"; vs fp32 reference {maxabs(got[v], ref[v]):.2e}; same-dtype oracle vs fp32 reference {maxabs(same[v], ref[v]):.2e}"
    print(msg)
    assert e_same < 4e-2                       # measured 2.57e-2 / 2.28e-2 (two decorrelated bf16 evaluations, see the module docstring)
    if ref is not None:
        assert maxabs(got[v], ref[v]) <= 1.15 * maxabs(same[v], ref[v]) + 2e-3      # measured 3.05e-2 vs 3.12e-2
        assert rel(got[v], ref[v]) <= 1.15 * rel(same[v], ref[v])


def test_output_hidden_states_vs_reference_fixture():
    """``output_hidden_states=True`` (valley_model.py:281-282, 324-330 -> HF LlamaModel's all_hidden_states: the embeddings,
    every layer's output, the last entry after the final norm) against the reference's own tuple (g10, tools/gen_goldens_r3.py):
    fp32 mode within 1e-4, the half-precision production path at its storage tolerance; same logits as without the flag."""
    g = np.load(os.path.join(GOLD, "g10_hidden_states.npz"))
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224).cuda()
    v = mask.astype(bool)[:, ::4]
    from valley_amd import runtime
    half = "fp16" if runtime.HALF == torch.float16 else "bf16"      # the production engines in the library's storage type
    for precision, tol in (("fp32", 1e-4), (half, 6e-2)):
        model = build_model("mean", precision)
        kw = dict(input_ids=torch.from_numpy(ids).cuda(), images=images, attention_mask=torch.from_numpy(mask).cuda())
        out = model(output_hidden_states=True, **kw)
        assert len(out.hidden_states) == int(g["n"]) == G.GCFG["L"] + 1
        errs = []
        for i, h in enumerate(out.hidden_states):
            assert tuple(h.shape) == (2, ids.shape[1], G.GCFG["H"])
            errs.append(maxabs(h.float().cpu().numpy()[:, ::4, ::2][v], g[f"hs{i}"][v]))
        plain = model(**kw)
        same = maxabs(out.logits.cpu().numpy(), plain.logits.cpu().numpy())
        print(f"{precision}: hidden_states max-abs vs the reference's {['%.2e' % e for e in errs]} (|h| up to 4.9), logits with/without the flag differ by {same:.1e}")
        assert max(errs) < tol
        assert maxabs(out.logits.cpu().numpy()[:, ::4][v], g["logits"][v]) < (TOL_FP32 if precision == "fp32" else 4.5e-2)
        assert same < (1e-5 if precision == "fp32" else 4.5e-2)       # the collecting pass flushes the residual per layer: rounding order only
        assert plain.hidden_states is None



def test_output_attentions_vs_reference_fixture():
    """``output_attentions=True`` (valley_model.py:281, 324-330 -> HF LlamaModel's all_self_attns, which HF's eager attention
    fills with softmax(QK^T / sqrt(d) + mask)) against the reference's own tuple (g11, tools/gen_goldens_r4.py; the CPU oracle is
    pinned to the same fixture in tests/test_oracle_golden.py): fp32 mode within 1e-4, the half-precision production path at
    its storage tolerance; padded keys exactly zero, valid rows sum to one; logits unchanged by the flag."""
    g = np.load(os.path.join(GOLD, "g11_attentions.npz"))
    T = G.GCFG["T"]
    ids, mask = G.golden_ids("main")
    images = torch.from_numpy(G.golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224).cuda()
    S = ids.shape[1]
    vq = mask.astype(bool)[:, ::4]
    from valley_amd import runtime
    half = "fp16" if runtime.HALF == torch.float16 else "bf16"      # the production engines in the library's storage type
    for precision, tol in (("fp32", 1e-4), (half, 2e-2)):
        model = build_model("mean", precision)
        kw = dict(input_ids=torch.from_numpy(ids).cuda(), images=images, attention_mask=torch.from_numpy(mask).cuda())
        out = model(output_attentions=True, **kw)
        assert len(out.attentions) == int(g["n"]) == G.GCFG["L"]
        errs = []
        for i, a in enumerate(out.attentions):
            assert tuple(a.shape) == (2, G.GCFG["heads"], S, S)
            got = a.float().cpu().numpy()
            for b in range(2):
                rows = got[b][:, ::4][:, vq[b]]
                errs.append(maxabs(rows, g[f"attn{i}"][b][:, vq[b]]))
                assert np.all(rows[..., ~mask.astype(bool)[b]] == 0)
                assert np.abs(rows.sum(-1) - 1).max() < (1e-5 if precision == "fp32" else 2e-2)
        plain = model(**kw)
        same = maxabs(out.logits.cpu().numpy(), plain.logits.cpu().numpy())
        print(f"{precision}: attention probabilities max-abs vs the reference's {['%.2e' % e for e in errs]}, logits with/without the flag differ by {same:.1e}")
        assert max(errs) < tol
        assert same < (1e-5 if precision == "fp32" else 4.5e-2) and plain.attentions is None
