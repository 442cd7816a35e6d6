"""Host-side splice index logic (valley_amd/splice.py) against the oracle's statement-for-statement
restatement of valley_model.py:195-247 and the reference-captured fixtures — bit-exact, since it is
integer work: applying the row map to (embedding rows, visual rows) must reproduce ``embeds``."""
import os

import numpy as np
import pytest
import torch

from oracle import valley_oracle as O
from tests import golden_cfg as G
from valley_amd.splice import build_row_map

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def tok():
    return O.TokenIds(**G.special())


def apply_map(row_map, embed, visual):
    out = np.empty((row_map.size, embed.shape[1]), np.float32)
    for i, v in enumerate(row_map):
        out[i] = embed[v] if v >= 0 else visual[-v - 1]
    return out


@pytest.mark.parametrize("case,Ts", [("main", [4, 4]), ("mixed", [4]), ("two_images", [4]), ("frame_mismatch", [4]),
                                     ("list", [2, 3]), ("decode", [4])])
def test_row_map_reproduces_oracle_splice(case, Ts):
    ids, _ = G.golden_ids(case)
    H = 8
    rng = np.random.default_rng(0)
    embed = rng.standard_normal((G.GCFG["vocab"], H)).astype(np.float32)
    feats = [torch.from_numpy(rng.standard_normal((t, 257, H)).astype(np.float32)) for t in Ts]
    ref = O.splice_visual_tokens(torch.from_numpy(ids), torch.from_numpy(embed[ids]), feats, tok()).numpy()
    visual = np.concatenate([np.concatenate([f[:, 1:].mean(0).numpy(), f[:, 0].numpy()], 0) for f in feats], 0)
    rm = build_row_map(ids, Ts, tok())
    got = apply_map(rm, embed, visual).reshape(ref.shape)
    assert np.array_equal(got, ref)


def test_row_map_matches_reference_fixture_positions():
    """On the reference-captured ``embeds`` the rows the map marks as visual must differ from the plain
    embedding rows and the others must equal them exactly."""
    g = np.load(os.path.join(GOLD, "g2_forward_mean.npz"))
    ids = g["ids"]
    emb_tab = G.llama_state()["model.embed_tokens.weight"]
    rm = build_row_map(ids, [4, 4], tok()).reshape(ids.shape)
    plain = emb_tab[ids]
    same = np.all(g["embeds"] == plain, axis=-1)
    assert np.array_equal(same, rm >= 0)


def test_errors_match_reference_messages():
    g = np.load(os.path.join(GOLD, "g3_errors.npz"))
    for case in ("cut", "unbalanced"):
        ids, _ = G.golden_ids(case)
        with pytest.raises(ValueError) as e:
            build_row_map(ids, [4], tok())
        assert str(g[case]) == f"ValueError: {e.value}"


def test_edge_cases():
    t = tok()
    # text-only batch: identity map, clip list untouched
    ids = np.arange(3, 23, dtype=np.int64).reshape(2, 10)
    assert np.array_equal(build_row_map(ids, [], t), ids.reshape(-1).astype(np.int32))
    # multimodal sample but no clip supplied -> IndexError like image_features[cur_image_idx]
    ids, _ = G.golden_ids("decode")
    with pytest.raises(IndexError):
        build_row_map(ids, [], t)
    # prompt truncated right after the patches: the unguarded index of the reference
    short = np.asarray([[1, t.im_start_token] + [t.im_patch_token] * 200 + [t.im_end_token]], dtype=np.int64)
    with pytest.raises(IndexError):
        build_row_map(short, [4], t)
    with pytest.raises(IndexError):
        O.splice_visual_tokens(torch.from_numpy(short), torch.zeros(1, short.shape[1], 4),
                               [torch.zeros(4, 257, 4)], t)
