"""GPU parity at PRODUCTION shapes (BASELINE.json configs[1]/[2]/[4]: 7B and 13B Llama layer shapes, the hot GEMM
shapes of the c2 / c3 steps) against the CPU oracle (oracle/valley_oracle.py, fp32) or an fp32 product.

Reference math: hf:llama/modeling_llama.py:217-332 as called from /root/reference/valley/model/valley_model.py:249-254
(decoder), :304-305 (lm_head), valley/serve/model_worker.py:380-387 (one-token KV step).  The oracle needs seconds on
the host at 2 layers; layer count does not change any kernel shape, so 2 layers exercise exactly the launches of the
32- / 40-layer models.

Tolerances (stated; bf16 GEMM operands, fp32 accumulation, fp32 residual stream, vs fp32 reference), measured on MI355X
and asserted at ~1.5x the measurement — see the prints:
    hidden state / logits after 2 layers at 7B | 13B shapes   rel-L2 1.10e-2 | 1.42e-2 measured, < 1.7e-2 | 2.2e-2 asserted,
        AND <= 1.15x the error of the oracle's own bf16-storage evaluation (oracle.rounding(): 1.12e-2 | 1.44e-2): the HIP
        path adds nothing to what storing GEMM operands in bf16 costs at these widths; the fp32 mode
        (tests/test_precise_gpu.py) is at 7e-6 on the same problem
    bf16-output GEMM vs fp32 product              rel-L2 < 4e-3 (one bf16 rounding of the result = 2^-9 relative rms)
    fp32-output GEMM vs fp32 product              max-abs < 2e-4 sqrt(K) (summation order only: bf16 products are exact)
"""
import math

import numpy as np
import pytest
import torch

from valley_amd.runtime import HALF  # the library's 16-bit storage type: bf16, or fp16 under VALLEY_PRECISION=fp16 (this process is bound by the environment)

pytestmark = pytest.mark.gpu

SHAPES = {"7b": dict(H=4096, heads=32, I=11008, eps=1e-5, S=328), "13b": dict(H=5120, heads=40, I=13824, eps=1e-6, S=336)}
VOCAB = 512


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


_STATE = {}
_ORACLE = {}


def _llama(name, layers=2):
    """(HipLlama on the device, state dict, oracle cfg) for a 2-layer model at the named shapes (cached per session:
    generating 0.6 G deterministic fp32 numbers takes several seconds)."""
    from oracle import valley_oracle as O
    from valley_amd import weights as W
    from valley_amd.llama import HipLlama
    if name not in _STATE:
        s = SHAPES[name]
        sd = W.valley_llama_weights(5, VOCAB, s["H"], s["I"], layers)
        ll = HipLlama(s["H"], s["heads"], s["I"], layers, VOCAB, s["eps"]).load_state_dict(sd)
        cfg = O.LlamaCfg(hidden=s["H"], heads=s["heads"], intermediate=s["I"], layers=layers, vocab=VOCAB, eps=s["eps"])
        _STATE[name] = (ll, sd, cfg)
    return _STATE[name]


@pytest.mark.parametrize("name", ["7b", "13b"])
@pytest.mark.parametrize("mode", ["tiles", "tuned"])
def test_llama_layers_vs_oracle(name, mode, monkeypatch):
    """Prefill of B=2 sequences (the second left-padded by 11) through 2 decoder layers + final norm + lm_head at 7B /
    13B shapes, whole-tile and tuned (shipped table / split-K pairs / packed weights) dispatch, vs oracle.llama_forward;
    then ONE hipGraph-captured DecodeSession step on the same cache vs the oracle's KV step."""
    from oracle import valley_oracle as O
    from valley_amd import ops, weights as W
    from valley_amd.decode import DecodeSession
    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    ll, sd, cfg = _llama(name)
    s = SHAPES[name]
    B, S, H = 2, s["S"], s["H"]
    emb = W.det_normal(9, f"emb.{name}", (B, S, H), 0.5)
    mask = np.ones((B, S), np.int64)
    mask[1, :11] = 0
    cache = ll.new_cache(B, S + 8)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    cache.key_valid[:, :S] = torch.from_numpy(mask).to(torch.uint8).cuda()
    passes = 1
    if mode == "tuned":
        passes = 260                                               # let the online tuner decide any shape not in the table
    for it in range(passes):
        cache.seq_len = 0
        x = ll.forward(torch.from_numpy(emb).cuda().view(B * S, H).clone(), B, S, cache)
        torch.cuda.synchronize()
        if ops.tuning_pending() == 0:
            break
    got_h = x.float().view(B, S, H).cpu().numpy()
    got_l = ll.logits(x).view(B, S, -1).cpu().numpy()
    v = mask.astype(bool)
    if name not in _ORACLE:                                      # CPU oracle, fp32 and bf16-storage evaluation (shared by both modes)
        with torch.no_grad():
            lm = torch.from_numpy(sd["lm_head.weight"])
            ref_h, past = O.llama_forward(torch.from_numpy(emb), sd, cfg, torch.from_numpy(mask))
            ref_l = torch.nn.functional.linear(ref_h, lm)
            with O.rounding():
                same_h, _ = O.llama_forward(torch.from_numpy(emb), sd, cfg, torch.from_numpy(mask))
                same_l = torch.nn.functional.linear(same_h, lm.to(HALF).float())
        _ORACLE[name] = (ref_h, ref_l, past, rel(same_h.numpy()[v], ref_h.numpy()[v]), rel(same_l.numpy()[v], ref_l.numpy()[v]))
    ref_h, ref_l, past, sh, sl = _ORACLE[name]
    rh, rl, ml = rel(got_h[v], ref_h.numpy()[v]), rel(got_l[v], ref_l.numpy()[v]), maxabs(got_l[v], ref_l.numpy()[v])
    print(f"{name}/{mode}: hidden rel-L2 {rh:.2e}  logits rel-L2 {rl:.2e} max-abs {ml:.3e} (|logit| max "
          f"{float(np.abs(ref_l.numpy()[v]).max()):.2f}); the oracle's bf16-storage evaluation: hidden {sh:.2e} logits {sl:.2e}")
    bound = {"7b": 1.7e-2, "13b": 2.2e-2}[name]
    assert rh < bound and rl < bound
    assert rh <= 1.15 * sh and rl <= 1.15 * sl                  # no error beyond what bf16 storage itself costs
    assert ops.sk_error_flag("cuda:0") == 0
    # ---- one decode step (GEMV kernels, fused RoPE/append/attention) on the prefilled cache
    tok = torch.tensor([3, 7], dtype=torch.long)
    sess = DecodeSession(ll, cache, use_graph=True)
    sess.begin(tok.cuda())
    sess.step()
    torch.cuda.synchronize()
    got_d = sess.logits[:, :VOCAB].cpu().numpy()
    with torch.no_grad():
        e1 = torch.from_numpy(sd["model.embed_tokens.weight"])[tok][:, None]
        # the device embedding table is bf16: feed the oracle the same rounded rows (the table is a weight, not arithmetic)
        e1 = e1.to(HALF).float()
        m1 = torch.cat([torch.from_numpy(mask), torch.ones((B, 1), dtype=torch.long)], 1)
        h1, _ = O.llama_forward(e1, sd, cfg, m1, past=past)
        ref_d = torch.nn.functional.linear(h1, torch.from_numpy(sd["lm_head.weight"]))[:, 0].numpy()
    print(f"{name}/{mode}: decode-step logits rel-L2 {rel(got_d, ref_d):.2e} max-abs {maxabs(got_d, ref_d):.3e}")
    assert rel(got_d, ref_d) < {"7b": 1.7e-2, "13b": 2.2e-2}[name]


# ---- the GEMM shapes that carry the c2 / c3 steps ------------------------------------------------
HOT = [  # M, N, K, epilogue, bias, label
    (1312, 22016, 4096, 2, False, "c2 gate|up SwiGLU"),
    (1312, 12288, 4096, 0, False, "c2 q|k|v"),
    (1312, 4096, 11008, 0, False, "c2 down"),
    (1312, 4096, 4096, 0, False, "c2 o"),
    (2688, 27648, 5120, 2, False, "c3 gate|up SwiGLU"),
    (2688, 15360, 5120, 0, False, "c3 q|k|v"),
    (2688, 5120, 13824, 0, False, "c3 down"),
    (2688, 5120, 5120, 0, False, "c3 o"),
    (8224, 4096, 1024, 1, True, "ViT fc1 quick_gelu"),
    (8224, 3072, 1024, 0, True, "ViT q|k|v"),
    (8224, 1024, 4096, 0, True, "ViT fc2"),
    (8224, 1024, 1024, 0, True, "ViT out"),
    (2688, 32008, 5120, 0, False, "c3 lm_head"),
    # configs[3]: the per-GPU shapes at N = 1 (8 clips x 32 frames, S = 352: M = 2816) and the REPLICATED prefill of all
    # 64 sequences (M = 64 x 352 = 22528) — VERDICT r3 item 1a
    (2816, 27648, 5120, 2, False, "c4 gate|up SwiGLU"),
    (2816, 15360, 5120, 0, False, "c4 q|k|v"),
    (2816, 5120, 13824, 0, False, "c4 down"),
    (2816, 5120, 5120, 0, False, "c4 o"),
    (22528, 27648, 5120, 2, False, "c4x8 gate|up SwiGLU"),
    (22528, 15360, 5120, 0, False, "c4x8 q|k|v"),
    (22528, 5120, 13824, 0, False, "c4x8 down"),
    (22528, 5120, 5120, 0, False, "c4x8 o"),
]


def _ref_product(a, w, bias, epi):
    """fp32 product on the device in row chunks (plain PyTorch fp32 matmul: the checker, not the product path)."""
    outs = []
    wf = w.float()
    for r0 in range(0, a.shape[0], 512):
        c = a[r0:r0 + 512].float() @ wf.t()
        if bias is not None:
            c = c + bias
        if epi == 1:
            c = c * torch.sigmoid(1.702 * c)
        elif epi == 2:
            c = torch.nn.functional.silu(c[:, 0::2]) * c[:, 1::2]
        outs.append(c)
    return torch.cat(outs, 0)


def _relerr(got, ref):
    return float((got.float() - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("M,N,K,epi,has_bias,label", HOT, ids=[h[5].replace(" ", "_").replace("|", "") for h in HOT])
def test_gemm_hot_shapes_vs_fp32(M, N, K, epi, has_bias, label, monkeypatch):
    from valley_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g, device=d).to(HALF)
    w = (torch.randn((N, K), generator=g, device=d) * 0.05).to(HALF)
    # transpose detector: one hot row / column pair (symmetric random data would hide a swapped fragment)
    a[M - 3, :] = 0
    a[M - 3, K - 5] = 2.0
    bias = (torch.randn((N,), generator=g, device=d) * 0.5) if has_bias else None
    ref = _ref_product(a, w, bias, epi)
    No = ref.shape[1]
    # 1. whole-tile kernel, static heuristic (bf16 out) and fp32 out where the path uses it
    out = ops.gemm_mfma(a, w, bias, epilogue=epi)
    e_tile = _relerr(out, ref)
    assert e_tile < 4e-3, (label, e_tile)
    if epi == 0:
        o32 = ops.gemm_mfma(a, w, bias, out_dtype=torch.float32)
        assert float((o32 - ref).abs().max()) < 2e-4 * math.sqrt(K), label
    # 2. what the step really runs: the tuned dispatch with the shipped table, on the block-ordered weight copy
    #    for the Llama projections (ops.PackedWeight), through gemm2 for the split-K pairs
    monkeypatch.setattr(ops, "GEMM_MODE", "tuned")
    wp = ops.PackedWeight(w) if (M in (1312, 2688, 2816, 22528) and N % 64 == 0) else w
    e_tuned = e_pair = None
    for _ in range(200):
        out = ops.gemm(a, wp, bias, epilogue=epi)
        torch.cuda.synchronize()
        e_tuned = _relerr(out, ref)
        assert e_tuned < 4e-3, (label, e_tuned)
        if ops.tuning_pending() == 0:
            break
    if epi == 0 and N in (1024, 4096, 5120):                      # the projections that feed a residual add
        o0 = torch.empty((M, N), dtype=HALF, device=d)
        o1 = torch.empty_like(o0)
        for _ in range(300):
            n = ops.gemm2(a, wp, o0, o1, bias)
            torch.cuda.synchronize()
            got = o0.float() + o1.float() if n == 2 else o0.float()
            e_pair = _relerr(got, ref)
            assert e_pair < 5e-3, (label, n, e_pair)             # two bf16 roundings of half-sized partials
            if ops.tuning_pending() == 0:
                break
    # 3. stream-K kernel
    e_sk = _relerr(ops.gemm_streamk(a, w, bias, epilogue=epi), ref)
    assert e_sk < 4e-3, (label, e_sk)
    assert ops.sk_error_flag(d) == 0
    print(f"{label}: {M}x{N}x{K} rel-L2 tile {e_tile:.2e} tuned {e_tuned:.2e} pair {e_pair} stream-K {e_sk:.2e}; out cols {No}")


# ---- the v2 / v3 temporal modules at production width ------------------------------------------------
@pytest.mark.parametrize("method", ["mean", "max", "temporal_importance", "temporal_transformer"])
def test_temporal_pooling_at_production_width_vs_oracle(method):
    """All four pooling variants at H = 4096 (the 7B projector width), T = 8 frames, B = 2 clips: the projected visual
    tokens [B, 256+T, H] of the HIP path (vly_pool_tokens / vly_temporal_scores / temporal_delta.hip + MFMA GEMMs) against
    the oracle (valley_model.py:187-193, 206-215, 113-121, 123-133), on a 3-layer ViT-L/14-geometry tower so that the
    oracle's tower finishes in seconds; prints the time of the pooling + projection stage."""
    from oracle import valley_oracle as O
    from tests import golden_cfg as G
    from valley_amd import valley_model as vm, weights as W
    H, T, B, V = 4096, 8, 2, 64
    cfg = vm.ValleyConfig(vocab_size=V, hidden_size=H, intermediate_size=11008, num_hidden_layers=0, num_attention_heads=32,
                          num_key_value_heads=32, rms_norm_eps=1e-5)
    cfg.use_mm_proj, cfg.mm_hidden_size, cfg.mm_vision_select_layer = True, 1024, -2
    sd = W.valley_llama_weights(31, V, H, 11008, 0)
    s = 31
    if method == "temporal_importance":
        sd["model.pooling_layer.weight"] = W.det_normal(s, "pool.w", (1, H * 256), 0.002)
        sd["model.pooling_layer.bias"] = W.det_normal(s, "pool.b", (1,), 0.01)
    if method == "temporal_transformer":
        shapes = {"self_attn.in_proj_weight": (3 * H, H), "self_attn.in_proj_bias": (3 * H,), "self_attn.out_proj.weight": (H, H),
                  "self_attn.out_proj.bias": (H,), "linear1.weight": (2048, H), "linear1.bias": (2048,), "linear2.weight": (H, 2048),
                  "linear2.bias": (H,), "norm1.weight": (H,), "norm1.bias": (H,), "norm2.weight": (H,), "norm2.bias": (H,)}
        for k, shp in shapes.items():
            t = W.det_normal(s, "tde." + k, shp, 0.02)
            sd["model.transformer_delta_encoder.layers.0." + k] = (t + 1.0 if k in ("norm1.weight", "norm2.weight") else t).astype(np.float32)
        sd["model.position_matrix"] = O.sinusoid_position_matrix(2048, H).numpy()
    model = vm.ValleyLlamaForCausalLM(cfg)
    model.load_state_dict(sd)
    vs = G.vision_state()
    tower = vm.build_vision_tower(dict(intermediate_size=G.GCFG["VI"], num_hidden_layers=G.GCFG["VL"]), state_dict=vs)
    mm = model.get_model()
    mm.vision_tower = tower
    mm.patch_pooling_method = method
    px = torch.from_numpy(G.golden_pixels(B * T, "n4")).view(B, T, 3, 224, 224)
    pooled, Ts = mm.encode_clips(px.cuda())
    vis = mm.project_pooled(pooled).float().view(B, 256 + T, H).cpu().numpy()
    with torch.no_grad():
        vcfg = O.VisionCfg(intermediate=G.GCFG["VI"], layers=G.GCFG["VL"])
        ref = []
        for b in range(B):
            proj = O.mm_project(O.vit_select(px[b], vs, vcfg, -2), sd)
            patches, cls = O.pool_clip(proj, method, sd)
            ref.append(torch.cat([patches, cls], 0))
        ref = torch.stack(ref).numpy()
    # time the stage after the tower (pool + project, or project + pool + v2 / v3 module)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    frames = px.cuda().view(-1, 3, 224, 224)
    for _ in range(2):
        ev[0].record()
        tower.encode(frames, -2)
        ev[1].record()
        ev[2].record()
        mm.project_pooled(mm.encode_clips(px.cuda())[0])
        ev[3].record()
    torch.cuda.synchronize()
    e = rel(vis, ref)
    print(f"N4 {method}: visual tokens [B={B}, {256 + T}, H={H}] rel-L2 {e:.2e} max-abs {maxabs(vis, ref):.3e} "
          f"(|ref| max {float(np.abs(ref).max()):.2f}); pooling+projection stage {ev[2].elapsed_time(ev[3]) - ev[0].elapsed_time(ev[1]):.3f} ms")
    assert e < 8e-3


# ---- BASELINE.json configs[3]: the replicated prefill of all 64 sequences ---------------------------------------------------
def test_c4_replicated_prefill_64_sequences_vs_oracle(monkeypatch):
    """configs[3]'s LLM step as every rank runs it: B = 64 sequences x S = 352 through 2 decoder layers at the 13B shapes
    (M = 22528 rows per GEMM, tuned dispatch with the shipped M = 22528 decisions) + lm_head, into a KV cache whose per-layer
    K / V tensors are > 4 GB each (ctx_max 6656: 64 x 40 x 6656 x 256 B = 4.36 GB) — the size at which round 3's LDS-DMA
    attention kernel fell back to the register-staged one; its descriptors now start at the (batch, head) slice.  Sequences
    are independent (valley_model.py:249-254 -> hf LlamaModel.forward), so the oracle evaluates a SLICE of them — rows 0, 31
    (left-padded by 9) and 63 — and the HIP rows are held to it; attention of BOTH kernels on the big cache is compared too."""
    from oracle import valley_oracle as O
    from valley_amd import ops, weights as W
    monkeypatch.setattr(ops, "GEMM_MODE", "tuned")
    ll, sd, cfg = _llama("13b")
    B, S, H, CTX = 64, 352, 5120, 6656
    rows = [0, 31, 63]
    emb = W.det_normal(19, "emb.c4", (B, S, H), 0.5)
    mask = np.ones((B, S), np.int64)
    mask[31, :9] = 0
    cache = ll.new_cache(B, CTX)
    assert cache.k[0].numel() * 2 > (1 << 32)                      # the > 4 GB case itself
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    cache.key_valid[:, :S] = torch.from_numpy(mask).to(torch.uint8).cuda()
    x0 = torch.from_numpy(emb).cuda().view(B * S, H)
    for it in range(260):                                            # (the shipped table decides these shapes: one pass)
        cache.seq_len = 0
        x = ll.forward(x0.clone(), B, S, cache)
        torch.cuda.synchronize()
        if ops.tuning_pending() == 0:
            break
    got_h = x.float().view(B, S, H)[rows].cpu().numpy()
    got_l = ll.logits(x).view(B, S, -1)[rows].cpu().numpy()
    with torch.no_grad():
        ref_h, _ = O.llama_forward(torch.from_numpy(emb[rows]), sd, cfg, torch.from_numpy(mask[rows]))
        ref_l = torch.nn.functional.linear(ref_h, torch.from_numpy(sd["lm_head.weight"]))
    v = mask[rows].astype(bool)
    rh, rl = rel(got_h[v], ref_h.numpy()[v]), rel(got_l[v], ref_l.numpy()[v])
    print(f"c4 replicated prefill B=64 S=352 (M=22528), cache {cache.k[0].numel() * 2 / 2**30:.2f} GB per tensor: rows {rows} "
          f"hidden rel-L2 {rh:.2e} logits rel-L2 {rl:.2e} max-abs {maxabs(got_l[v], ref_l.numpy()[v]):.3e}")
    assert rh < 2.2e-2 and rl < 2.2e-2                                # the 13B-shape bound of test_llama_layers_vs_oracle
    assert ops.sk_error_flag("cuda:0") == 0
    # the attention call of that step on the > 4 GB cache: LDS-DMA kernel (default) == register-staged kernel (bit-identical)
    import os, subprocess, sys
    qkv = (torch.randn((B * S, 3 * H), device="cuda") * 0.5).to(HALF)
    a2 = ops.llama_attention(qkv, cache.k[1], cache.v[1], cache.key_valid, B, S, ll.heads, 0)
    torch.cuda.synchronize()
    code = ("import torch,sys; sys.path.insert(0, %r); from valley_amd import ops; from valley_amd.runtime import HALF; "
            "B,S,H,CTX=64,352,5120,6656; torch.manual_seed(3); "
            "k=(torch.randn((B,40,CTX,128),device='cuda')*0.5).to(HALF); v=(torch.randn((B,40,CTX,128),device='cuda')*0.5).to(HALF); "
            "q=(torch.randn((B*S,3*H),device='cuda')*0.5).to(HALF); kv=torch.ones((B,CTX),dtype=torch.uint8,device='cuda'); kv[5,:17]=0; "
            "o=ops.llama_attention(q,k,v,kv,B,S,40,0); torch.cuda.synchronize(); print('SUM', float(o.float().abs().sum()), float(o.float()[-1].sum()))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for ver in ("2", "1"):                                           # the env switch is read once per process: two children
        env = dict(os.environ, VLY_LLAMA_ATTN=ver, VALLEY_EXPERIMENTAL="1")     # (the register-staged witness kernel: experimental library)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("SUM")][-1])
    print("attention on a 4.36 GB cache, llama_attn2 vs llama_attn:", outs)
    assert outs[0] == outs[1]
    assert torch.isfinite(a2.float()).all()


# ---- BASELINE.json configs[2] end to end --------------------------------------------------------------------------------
def test_c3_shape_end_to_end_vs_oracle():
    """configs[2] as a whole — B = 8 clips x T = 16 frames (128 frames through the full-depth ViT-L/14, 23 contributing
    layers) -> mean pool + projector -> splice into 8 prompts of S = 336 -> decoder at the Vicuna-13B layer shapes
    (H 5120, 40 heads, I 13824, eps 1e-6) -> lm_head on every position — through the reference-shaped API
    (ValleyLlamaForCausalLM.forward), tuned dispatch, against the CPU oracle on the same deterministic weights and
    pixels.  Only the decoder DEPTH is reduced (2 of 40 layers: depth changes no kernel shape) so that the oracle needs
    ~30 s of host time instead of ten minutes."""
    from oracle import valley_oracle as O
    from valley_amd import ops, valley_model as vm, weights as W
    B, T, H, heads, I, V, VT = 8, 16, 5120, 40, 13824, 512, 500
    S = 320 + T
    cfg = vm.ValleyConfig(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=2, num_attention_heads=heads,
                          num_key_value_heads=heads, rms_norm_eps=1e-6, max_position_embeddings=2048)
    cfg.use_mm_proj, cfg.mm_hidden_size, cfg.mm_vision_select_layer = True, 1024, -2
    _, sd, lcfg = _llama("13b")
    vsd = W.clip_vision_weights(3, layers=24)
    model = vm.ValleyLlamaForCausalLM(cfg)
    model.load_state_dict(sd)
    tower = vm.build_vision_tower(None, state_dict=vsd)
    special = W.SPECIAL_IDS(VT)
    for k, v in special.items():
        setattr(tower.config, k, v)
    model.get_model().vision_tower = tower
    ids = np.stack([W.synthetic_prompt(7 + b, T, VT, ids=special) for b in range(B)])
    assert ids.shape == (B, S)
    px = torch.from_numpy(W.det_normal(4, "px.c3", (B, T, 3, 224, 224)))
    saved_mode = ops.GEMM_MODE
    ops.GEMM_MODE = "tuned"
    try:
        for _ in range(40):                                  # let the online tuner settle any shape not in the shipped table
            out = model(input_ids=torch.from_numpy(ids).cuda(), images=px.cuda())
            torch.cuda.synchronize()
            if ops.tuning_pending() == 0:
                break
    finally:
        ops.GEMM_MODE = saved_mode
    got = out.logits.cpu().numpy()
    assert got.shape == (B, S, V) and ops.sk_error_flag("cuda:0") == 0
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(16, nthr))                     # torch's CPU GEMM peaks at 16 threads on the GPU box's host
    try:
        with torch.no_grad():
            ref, _, emb = O.valley_forward(torch.from_numpy(ids), px, sd, vsd, lcfg, O.VisionCfg(), O.TokenIds(**special))
    finally:
        torch.set_num_threads(nthr)
    ref = ref.numpy()
    e_l, m_l = rel(got, ref), maxabs(got, ref)
    got_emb = model.get_model().embed_inputs(torch.from_numpy(ids), px.cuda()).view(B, S, H).cpu().numpy()
    e_e = rel(got_emb, emb.numpy())
    print(f"c3 end to end (2 of 40 decoder layers): spliced embeddings rel-L2 {e_e:.2e}; logits rel-L2 {e_l:.2e} max-abs {m_l:.3e} "
          f"(|logit| max {float(np.abs(ref).max()):.2f})")
    assert e_e < 8e-3 and e_l < 2.2e-2


def test_c4_shape_tower_chunk_invariance():
    """The per-GPU tower shape of configs[3] (8 clips x 32 frames = 256 frames in one pass): with the batch-invariant
    whole-tile dispatch the 256-frame pass equals, bit for bit, the same frames encoded as 2 x 128 and as 8 x 32 — the
    encode is per-frame, however many frames share a launch (and whatever the row split does to the launches in tuned
    mode stays within fp32 summation-order noise)."""
    from valley_amd import ops, valley_model as vm
    tower = vm.build_vision_tower(None)
    tower.init_random(seed=2)
    g = torch.Generator(device="cuda").manual_seed(6)
    frames = torch.randn((256, 3, 224, 224), generator=g, device="cuda")
    a = tower.encode(frames, -2, chunk=256)
    assert torch.equal(a, tower.encode(frames, -2, chunk=128)) and torch.equal(a[:64], tower.encode(frames[:64], -2, chunk=32))
    assert torch.isfinite(a).all()
    saved = ops.GEMM_MODE
    ops.GEMM_MODE = "tuned"
    try:
        for _ in range(30):
            b = tower.encode(frames, -2)
            torch.cuda.synchronize()
            if ops.tuning_pending() == 0:
                break
    finally:
        ops.GEMM_MODE = saved
    r = float((b - a).norm() / a.norm())
    print("c4 tower, tuned (row-split + skinny remainder) vs whole tiles: rel-L2", r)
    assert r < 5e-3 and ops.sk_error_flag("cuda:0") == 0


def test_vit_two_stream_remainder_schedule_is_bit_identical(monkeypatch):
    """vision_tower.layer_forward_2s (the remainder rows of the row-split GEMMs on a side stream, forked after attention and
    joined before the next q|k|v GEMM) runs the same kernels on the same rows as the single-stream schedule: once the
    tuner has decided every shape, 32 frames through 4 ViT-L/14 layers give bit-identical features, repeatedly (a missing
    dependency between the streams would show as a difference or as run-to-run noise)."""
    from valley_amd import ops, vision_tower as vt
    from valley_amd import valley_model as vm
    monkeypatch.setattr(ops, "GEMM_MODE", "tuned")
    monkeypatch.setattr(ops, "ROW_SPLIT_MIN", 8192)          # F = 32: M = 8224 = 8192 + 32 remainder rows
    tower = vm.build_vision_tower(None)
    tower.init_random(seed=3, layers=4)
    g = torch.Generator(device="cuda").manual_seed(8)
    frames = torch.randn((32, 3, 224, 224), generator=g, device="cuda").to(HALF)
    assert ops.row_split(32 * 257) == 8192
    before = set(ops._ONLINE)                                  # shapes other tests of this process left undecided do not recur here
    for it in range(400):                                      # let the online tuner settle on this tower's shapes
        monkeypatch.setattr(vt, "TWO_STREAM", bool(it & 1))
        tower.encode(frames, select_layer=4)
        torch.cuda.synchronize()
        if set(ops._ONLINE) <= before and it >= 3:
            break
    assert set(ops._ONLINE) <= before
    monkeypatch.setattr(vt, "TWO_STREAM", False)
    ref = tower.encode(frames, select_layer=4).clone()
    monkeypatch.setattr(vt, "TWO_STREAM", True)
    for _ in range(5):
        got = tower.encode(frames, select_layer=4)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0.1
