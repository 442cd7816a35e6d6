"""vly_decode_attention_merged (round 4: the last of a head's four split workgroups merges the partials inside the attention
launch; the o projection is a plain GEMV) against round 3's form (vly_decode_attention_split + the merge in the prologue of
vly_gemv_attnmerge_bf16, held to the CPU oracle in tests/test_scale_gpu.py / test_depth_gpu.py): the same arithmetic in the same
order, so the bar is BIT-identity of every step's residual stream, logits, token and appended K / V rows.  A partial read before
its writer's write-through stores landed (the hand-off is a ticket counter, no fence) or a counter that is not back at zero
after a launch shows up as a different bit pattern / a hang-free wrong merge.
Reference path: serve/model_worker.py:380-394 (one-token forward per step) -> hf LlamaAttention.forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from valley_amd import lib as _vlib  # noqa: E402

if not _vlib.EXPERIMENTAL:
    # round 3's form (vly_decode_attention_split + vly_gemv_attnmerge_bf16) is the experimental library's: this module runs in the child
    # process tests/test_experimental_gpu.py starts with VALLEY_EXPERIMENTAL=1
    pytest.skip("round-3 split / merge form: experimental library only (tests/test_experimental_gpu.py runs this module)", allow_module_level=True)

from tests.test_scale_gpu import SHAPES, _llama  # noqa: E402


def _run(name, B, S, steps, monkeypatch, mode, graph=True, pad=0, per_row=False, extra=72):
    from valley_amd import decode, weights as W
    ll, sd, cfg = _llama(name)
    H = SHAPES[name]["H"]
    emb = torch.from_numpy(W.det_normal(37, f"emb.mg.{name}.{B}.{S}", (B, S, H), 0.5)).cuda()
    monkeypatch.setattr(decode, "MERGE_IN", mode)
    cache = ll.new_cache(B, S + extra)
    cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device="cuda")
    if pad:
        cache.key_valid[B - 1, :pad] = 0
    cache.seq_len = 0
    ll.forward(emb.view(B * S, H).clone(), B, S, cache)
    sess = decode.DecodeSession(ll, cache, use_graph=graph, per_row_positions=per_row)
    if per_row:
        sess.pos.copy_(torch.tensor([S, max(S - 119, 1)][:B], dtype=torch.int32))
        sess.tok.copy_(torch.tensor([9, 4][:B], dtype=torch.int32))
        sess.begin()
    else:
        sess.begin(torch.tensor([3, 7][:B], dtype=torch.long).cuda())
    out = []
    for _ in range(steps):
        t = sess.step().clone()
        out.append((sess.h.clone(), sess.logits.clone(), t))
    torch.cuda.synchronize()
    assert int(sess.arrivals.abs().sum().item()) == 0, "the ticket counters are not back at zero"
    return ll, out, cache


@pytest.mark.parametrize("name,B,S,pad,graph", [("13b", 1, 328, 0, True), ("13b", 2, 336, 11, True), ("7b", 2, 200, 7, True),
                                                ("13b", 1, 255, 0, False), ("13b", 1, 1020, 0, False), ("7b", 1, 3, 0, True)])
def test_merge_in_attention_is_bit_identical_to_merge_in_o_proj(name, B, S, pad, graph, monkeypatch):
    """configs[4]'s shape and the 7B class; batch 2 with a left-padded row; key-pass boundaries (255 -> 256 keys, 1020 -> two
    passes per split); a 3-token context (three of the four splits are EMPTY and still draw their ticket); 64 steps each."""
    steps = 64 if graph else 12
    ll, a, ca = _run(name, B, S, steps, monkeypatch, "oproj", graph, pad)
    _, b, cb = _run(name, B, S, steps, monkeypatch, "attn", graph, pad)
    for i in range(steps):
        assert torch.equal(a[i][0], b[i][0]), f"step {i}: residual stream differs"
        assert torch.equal(a[i][1], b[i][1]) and torch.equal(a[i][2], b[i][2]), f"step {i}: logits / token differ"
    for li in range(ll.L):
        assert torch.equal(ca.k[li], cb.k[li]) and torch.equal(ca.v[li], cb.v[li])


def test_merge_in_attention_with_per_row_positions(monkeypatch):
    ll, a, ca = _run("7b", 2, 300, 16, monkeypatch, "oproj", True, per_row=True)
    _, b, cb = _run("7b", 2, 300, 16, monkeypatch, "attn", True, per_row=True)
    for i in range(16):
        assert torch.equal(a[i][0], b[i][0]) and torch.equal(a[i][1], b[i][1]) and torch.equal(a[i][2], b[i][2]), i


def test_merged_entry_point_rejects_bad_arguments():
    from valley_amd import lib
    L = lib.load()
    x = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = x.data_ptr()
    # out missing / arrivals missing
    assert L.vly_decode_attention_merged(p, p, p, p, p, None, 0, p, None, p, 1, 1, 0, None, 0, 8, None) == -22
    assert L.vly_decode_attention_merged(p, p, p, p, p, None, 0, p, p, None, 1, 1, 0, None, 0, 8, None) == -22
