"""The persistent decode step (vly_decode_layers, valley_amd/csrc/decode_step.hip) against the five launches per layer it
replaces (vly_gemv_rmsnorm_bf16 / vly_decode_attention_split / vly_gemv_attnmerge_bf16 / vly_gemv_rmsnorm_bf16 / vly_gemv_bf16).

Reference path: serve/model_worker.py:380-394 (one-token forward per step) -> hf LlamaDecoderLayer.forward.  The launches are
held to the CPU oracle in tests/test_scale_gpu.py (2 layers at 7B / 13B shapes) and tests/test_depth_gpu.py (256 steps); the
persistent kernel reproduces their arithmetic operation for operation, so the bar here is BIT-IDENTITY, step after step:
residual stream, logits, chosen token, and every K / V row appended to the cache — over contexts that cross the 256-key pass
boundary of the split attention, left-padded rows, batch 2, per-row positions, and (one case) against the oracle itself.
Any stale cross-CU read inside the launch (the hand-offs go through write-through stores and sc1 loads, no fences) shows up
here as a different bit pattern."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from valley_amd import lib as _vlib  # noqa: E402

if not _vlib.EXPERIMENTAL:
    # vly_decode_layers is exported by libvalley_hip_exp.so only (include/valley_hip.h, EXPERIMENTAL section): this module runs in the
    # child process tests/test_experimental_gpu.py starts with VALLEY_EXPERIMENTAL=1
    pytest.skip("persistent decode step: experimental library only (tests/test_experimental_gpu.py runs this module)", allow_module_level=True)

from tests.test_scale_gpu import SHAPES, VOCAB, _llama, rel  # noqa: E402


def _sessions(name, B, S, extra, pad=0, per_row=False, graph=True, monkeypatch=None):
    """Two decode sessions on two identically prefilled caches: the launches and the persistent kernel."""
    from valley_amd import decode, weights as W
    ll, sd, cfg = _llama(name)
    H = SHAPES[name]["H"]
    emb = torch.from_numpy(W.det_normal(23, f"emb.dp.{name}.{B}.{S}", (B, S, H), 0.5)).cuda()
    out = []
    for persistent in (False, True):
        monkeypatch.setattr(decode, "PERSISTENT", persistent)
        cache = ll.new_cache(B, S + extra)
        cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
        if pad:
            cache.key_valid[B - 1, :pad] = 0
        cache.seq_len = 0
        ll.forward(emb.view(B * S, H).clone(), B, S, cache)
        sess = decode.DecodeSession(ll, cache, use_graph=graph, per_row_positions=per_row)
        assert sess.persistent == persistent
        out.append((sess, cache))
    return ll, out


def _compare_steps(ll, pair, steps, first_tok):
    (s0, c0), (s1, c1) = pair
    for s in (s0, s1):
        s.begin(first_tok.cuda())
    for i in range(steps):
        t0 = s0.step().clone()
        t1 = s1.step().clone()
        torch.cuda.synchronize()
        s1.check()
        assert torch.equal(s0.h, s1.h), f"step {i}: residual stream differs (max {float((s0.h - s1.h).abs().max()):.3e})"
        assert torch.equal(s0.logits, s1.logits), f"step {i}: logits differ"
        assert torch.equal(t0, t1), f"step {i}: token differs"
        assert torch.isfinite(s1.logits[:, :ll.V]).all()
    for li in range(ll.L):
        assert torch.equal(c0.k[li], c1.k[li]) and torch.equal(c0.v[li], c1.v[li]), f"layer {li}: appended K / V rows differ"


@pytest.mark.parametrize("name,B,S,pad", [("13b", 1, 328, 0), ("13b", 2, 336, 11), ("7b", 1, 328, 0), ("7b", 2, 200, 7)])
def test_persistent_step_is_bit_identical_to_the_launches(name, B, S, pad, monkeypatch):
    """configs[4]'s shape (13B, 328-token prefix) and the 7B class, batch 1 and 2 (second row left-padded), hipGraph replay:
    24 steps, every step's h / logits / token and the whole cache bit-identical."""
    ll, pair = _sessions(name, B, S, 40, pad=pad, monkeypatch=monkeypatch)
    _compare_steps(ll, pair, 24, torch.tensor([3, 7][:B], dtype=torch.long))


@pytest.mark.parametrize("S", [255, 336, 1020, 1290])
def test_persistent_step_across_key_pass_boundaries(S, monkeypatch):
    """kv_len 256 -> 263 (a split's range ends exactly at a 64-key boundary), 337 (the new position falls into the
    padding of another split's pass), 1021 -> 1028 (the quarter grows past 256 keys: two passes per split) and 1291 (two
    passes, ragged): eager launches (no graph), 8 steps."""
    ll, pair = _sessions("13b", 1, S, 16, graph=False, monkeypatch=monkeypatch)
    _compare_steps(ll, pair, 8, torch.tensor([5], dtype=torch.long))


def test_persistent_step_with_per_row_positions(monkeypatch):
    """The continuous batcher's captured step (one device-side position per row): two rows at different positions."""
    from valley_amd import decode
    ll, pair = _sessions("7b", 2, 300, 40, per_row=True, monkeypatch=monkeypatch)
    for sess, cache in pair:
        sess.pos.copy_(torch.tensor([300, 181], dtype=torch.int32))          # row 1 is 119 tokens behind row 0
        sess.tok.copy_(torch.tensor([9, 4], dtype=torch.int32))
        cache.key_valid[1, 181:] = 1
        sess.begin()
    (s0, c0), (s1, c1) = pair
    for i in range(12):
        s0.step()
        s1.step()
        torch.cuda.synchronize()
        s1.check()
        assert torch.equal(s0.h, s1.h) and torch.equal(s0.logits, s1.logits) and torch.equal(s0.tok, s1.tok), i
    for li in range(ll.L):
        assert torch.equal(c0.k[li], c1.k[li]) and torch.equal(c0.v[li], c1.v[li])


def test_persistent_step_vs_oracle(monkeypatch):
    """One persistent step at the 13B shapes directly against the CPU oracle's KV step (the bound of
    tests/test_scale_gpu.py::test_llama_layers_vs_oracle for the launches: rel-L2 < 2.2e-2 with bf16 storage)."""
    from oracle import valley_oracle as O
    from valley_amd import decode, weights as W
    monkeypatch.setattr(decode, "PERSISTENT", True)
    ll, sd, cfg = _llama("13b")
    B, S, H = 1, 337 - 1, 5120
    emb = W.det_normal(29, "emb.dp.oracle", (B, S, H), 0.5)
    cache = ll.new_cache(B, S + 8)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    cache.seq_len = 0
    ll.forward(torch.from_numpy(emb).cuda().view(B * S, H).clone(), B, S, cache)
    sess = decode.DecodeSession(ll, cache, use_graph=True)
    assert sess.persistent
    tok = torch.tensor([11], dtype=torch.long)
    sess.begin(tok.cuda())
    sess.step()
    torch.cuda.synchronize()
    sess.check()
    got = sess.logits[:, :VOCAB].cpu().numpy()
    with torch.no_grad():
        _, past = O.llama_forward(torch.from_numpy(emb), sd, cfg, torch.ones((B, S), dtype=torch.long))
        e1 = torch.from_numpy(sd["model.embed_tokens.weight"])[tok][:, None].to(torch.bfloat16).float()
        h1, _ = O.llama_forward(e1, sd, cfg, torch.ones((B, S + 1), dtype=torch.long), past=past)
        ref = torch.nn.functional.linear(h1, torch.from_numpy(sd["lm_head.weight"]))[:, 0].numpy()
    e = rel(got, ref)
    print(f"persistent decode step at kv_len 337 vs oracle: logits rel-L2 {e:.2e}")
    assert np.isfinite(got).all() and e < 2.2e-2
