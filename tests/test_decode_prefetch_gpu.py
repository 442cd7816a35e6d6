"""The decode step with its weight prefetch (valley_amd/decode.py PREFETCH: extra workgroups of the attention launch,
vly_decode_attention_split_pf — or vly_prefetch on a second stream beside it, fork / join captured in the hipGraph) against the
same step without it: the prefetch kernel writes
nothing, so every step's residual stream, logits, token and appended K / V rows must be BIT-identical — a missing join
(o-proj starting before the attention partials are complete) or a capture that drops the side branch shows up here.
Reference path: serve/model_worker.py:380-394 (one-token forward per step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_scale_gpu import SHAPES, _llama  # noqa: E402


@pytest.mark.parametrize("name,B,graph,mode", [("13b", 1, True, "o"), ("7b", 2, True, "o"), ("13b", 1, False, "side"), ("7b", 1, True, "side")])
def test_prefetch_leaves_the_step_bit_identical(name, B, graph, mode, monkeypatch):
    from valley_amd import decode, weights as W
    ll, sd, cfg = _llama(name)
    H, S = SHAPES[name]["H"], 328
    emb = torch.from_numpy(W.det_normal(31, f"emb.pf.{name}.{B}", (B, S, H), 0.5)).cuda()
    runs = []
    for pf in ("0", mode):
        monkeypatch.setattr(decode, "PREFETCH", pf)
        cache = ll.new_cache(B, S + 24)
        cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
        cache.seq_len = 0
        ll.forward(emb.view(B * S, H).clone(), B, S, cache)
        sess = decode.DecodeSession(ll, cache, use_graph=graph)
        assert (sess._side is not None) == (pf == "side")
        sess.begin(torch.tensor([3, 7][:B], dtype=torch.long).cuda())
        hs, toks = [], []
        for _ in range(16):
            toks.append(sess.step().clone())
            hs.append((sess.h.clone(), sess.logits.clone()))
        torch.cuda.synchronize()
        runs.append((hs, toks, cache))
    (h0, t0, c0), (h1, t1, c1) = runs
    for i in range(16):
        assert torch.equal(h0[i][0], h1[i][0]) and torch.equal(h0[i][1], h1[i][1]) and torch.equal(t0[i], t1[i]), f"step {i}"
    for li in range(ll.L):
        assert torch.equal(c0.k[li], c1.k[li]) and torch.equal(c0.v[li], c1.v[li])


def test_prefetch_rejects_bad_arguments():
    from valley_amd import lib
    L = lib.load()
    x = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    assert L.vly_prefetch(x.data_ptr() + 4, 256, 8, None) == -22        # not 16-byte aligned
    assert L.vly_prefetch(x.data_ptr(), 256, 0, None) == -22
    assert L.vly_prefetch(x.data_ptr(), 8, 8, None) == 0                # nothing to read
    assert L.vly_prefetch(x.data_ptr(), 1024, 8, None) == 0
    torch.cuda.synchronize()
