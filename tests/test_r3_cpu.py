"""Round-3 host-side arithmetic, restated in plain Python / numpy and pinned (no GPU):
 * the (max, sum, P·V) merge that vly_gemv_attnmerge_bf16 applies to vly_decode_attention_split's partials is exact;
 * the unit schedule of the persistent GEMM's split-K remainder round (gemm_p4_kernel<…, SK>, valley_amd/csrc/gemm_bf16.hip)
   covers every (tile, K tile) exactly once, puts a tile's slices on that tile's XCD, and lets every owner wait only for
   units of its own or an earlier round;
 * the launcher's choice of S for the shapes DESIGN.md quotes."""
import math

import numpy as np
import pytest

SPLITS = 4          # VLY_DECODE_SPLITS


def split_ranges(kv_len):
    """decode_split_kernel: 64-aligned quarters of the keys."""
    chunk = (((kv_len + SPLITS - 1) // SPLITS) + 63) & ~63
    return [(sp * chunk, min(sp * chunk + chunk, kv_len)) for sp in range(SPLITS)]


@pytest.mark.parametrize("kv_len", [1, 5, 64, 65, 337, 464, 593, 1101, 4097])
def test_split_attention_merge_is_exact(kv_len):
    rng = np.random.default_rng(kv_len)
    s = rng.normal(size=kv_len) * 3.0                       # log2-domain scores of one head
    v = rng.normal(size=(kv_len, 128))
    valid = rng.random(kv_len) > 0.1
    valid[-1] = True                                        # the new token is always attended
    s = np.where(valid, s, -1.0e30)
    p = np.exp2(s - s.max())
    want = (p[:, None] * v).sum(0) / p.sum()
    parts = []
    covered = np.zeros(kv_len, dtype=int)
    for lo, hi in split_ranges(kv_len):
        if lo >= hi:
            parts.append((-1.0e30, 0.0, np.zeros(128)))     # the neutral element the kernel writes for an empty range
            continue
        covered[lo:hi] += 1
        m = s[lo:hi].max()
        e = np.where(s[lo:hi] > 0.5 * -1.0e30, np.exp2(s[lo:hi] - m), 0.0)
        parts.append((m, e.sum(), (e[:, None] * v[lo:hi]).sum(0)))
    assert (covered == 1).all() and all(lo % 64 == 0 for lo, _ in split_ranges(kv_len))
    assert sum(1 for lo, hi in split_ranges(kv_len) if lo <= kv_len - 1 < hi) == 1      # exactly one owner of the new position
    mx = max(m for m, _, _ in parts)
    L = sum(l * 2.0 ** (m - mx) for m, l, _ in parts)
    O = sum(o * 2.0 ** (m - mx) for m, _, o in parts)
    assert np.allclose(O / L, want, rtol=1e-12, atol=1e-12)


def choose_slices(tiles, nk, cus=256, slab_cap=512, max_ways=8):
    """launch_p4<…, SK>: the S that packs S x rem8 units into the fewest rounds per slice."""
    rem = tiles % cus
    rem8 = (rem + 7) & ~7
    best, S_best = 1e30, 1
    for S in range(1, max_ways + 1):
        if rem == 0:
            break
        if S > 1 and ((S - 1) * rem8 > slab_cap or nk // S < 2 or (S - 1) * ((nk + S - 1) // S) >= nk):
            continue
        cost = ((S * rem8 + cus - 1) // cus) / S + 0.06 * (S - 1)
        if cost < best - 1e-6:
            best, S_best = cost, S
    return S_best


def schedule(tiles, nk, S, G=256):
    """Per workgroup, the segments (tile, k0, k1, kind, round) gemm_p4_kernel<…, SK> walks: pool units first, then whole tiles."""
    full, rem = tiles // G, tiles % G
    rem8 = (rem + 7) & ~7
    units = S * rem8
    nks = (nk + S - 1) // S
    out = {}
    for b in range(G):
        segs, u, rnd = [], b, 0
        while u < units:
            if u % rem8 < rem:
                sl, p = divmod(u, rem8)
                k0 = sl * nks
                segs.append((full * G + p, k0, min(nk, k0 + nks), "owner" if min(nk, k0 + nks) == nk else "contrib", rnd, u))
            u += G
            rnd += 1
        for r in range(full):
            segs.append((r * G + b, 0, nk, "whole", None, None))
        out[b] = segs
    return out, rem8


@pytest.mark.parametrize("M,N,K,bm", [(1312, 22016, 4096, 192), (1312, 12288, 4096, 192), (1312, 4096, 11008, 192),
                                     (2688, 27648, 5120, 224), (2688, 5120, 13824, 224), (771, 1000, 256, 192), (513, 520, 128, 192)])
def test_splitk_remainder_schedule(M, N, K, bm):
    tiles = math.ceil(M / bm) * math.ceil(N / 256)
    nk = K // 64
    S = choose_slices(tiles, nk)
    sched, rem8 = schedule(tiles, nk, S)
    cover = {}
    unit_round = {}
    for b, segs in sched.items():
        for tile, k0, k1, kind, rnd, u in segs:
            assert k1 > k0
            for k in range(k0, k1):
                cover[(tile, k)] = cover.get((tile, k), 0) + 1
            if u is not None:
                unit_round[u] = rnd
                assert (u % 256) & 7 == tile & 7            # the unit runs on its tile's XCD (workgroup id & 7 == tile & 7)
                assert u % 256 == b
    assert len(cover) == tiles * nk and set(cover.values()) == {1}
    # an owner's contributors are the units rem8, 2 rem8, ... below it: never in a LATER round, so nothing can deadlock
    for b, segs in sched.items():
        for tile, k0, k1, kind, rnd, u in segs:
            if kind == "owner" and S > 1:
                assert k0 > 0
                for f in range(1, S):
                    assert unit_round[u - f * rem8] <= rnd
            if kind == "contrib":
                assert u < (S - 1) * rem8                   # slab / flag index range of the contributors


def test_slices_chosen_for_the_quoted_shapes():
    # 7B gate|up on 192-row tiles: 7 x 86 = 602 tiles = 2 rounds + 90 -> two slices (measured best: 206 / 216 / 242 us at 2 / 4 / 8)
    assert choose_slices(7 * 86, 64) == 2
    # 13B gate|up on 224-row tiles: 12 x 108 = 1296 = 5 rounds + 16 tiles -> many thin slices
    assert choose_slices(12 * 108, 80) >= 4
    # 13B down on 224-row tiles: 240 tiles on 256 CUs: nothing to split
    assert choose_slices(12 * 20, 216) == 1
    # an exact number of rounds: no remainder round at all
    assert choose_slices(512, 16) == 1
