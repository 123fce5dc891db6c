"""opal_recut_plan (opal_amd/csrc/recut.cpp): where the cuts go when the decomposition follows the plasma -- the
reference's partition rule (GridDesign::balanced, grid/mod.rs:157-206) applied to the particles where they stand, held to
what one re-cut can move.  Host arithmetic only: runs without a GPU."""
import numpy as np
import pytest

import opal_amd as oa


def loads(cuts, cost):
    p = np.concatenate([[0.0], np.cumsum(cost)])
    return np.array([p[cuts[r + 1]] - p[cuts[r]] for r in range(len(cuts) - 1)])


def test_a_balanced_partition_stays():
    cost = np.full(800, 100.0)
    cuts = np.arange(0, 801, 100)
    new, before, after = oa.recut_plan(cuts, cost)
    assert np.array_equal(new, cuts) and before == pytest.approx(1.0) and after == pytest.approx(1.0)


def test_unlimited_moves_give_the_reference_partition():
    """With nothing to hold it back the plan is grid/mod.rs:170-176 on the particle counts: cut i at the cell where the
    running sum reaches i / N of the total (the nearer of the two cells either side)."""
    rng = np.random.default_rng(3)
    cost = rng.integers(0, 2000, 4000).astype(float)
    cost[:1000] = 0.0  # vacuum in front, as in hole_boring.yaml
    n = 8
    cuts = np.arange(0, 4001, 500)
    new, before, after = oa.recut_plan(cuts, cost, max_shift=100000, min_size=8)
    # (a cut may not leave the two subdomains it separates: iterate, as a run does check after check)
    for _ in range(12):
        new, _, after = oa.recut_plan(new, cost, max_shift=100000, min_size=8)
    p = np.concatenate([[0.0], np.cumsum(cost)])
    for i in range(1, n):
        target = p[-1] * i / n
        k = int(np.searchsorted(p, target, side="left"))
        assert abs(p[new[i]] - target) <= max(abs(p[k] - target), abs(p[k - 1] - target)) + 1e-9, i
    assert before > 1.3 and after < 1.02
    assert np.array_equal(np.sort(new), new) and new[0] == 0 and new[-1] == 4000


def test_a_cut_is_held_to_what_one_recut_can_move():
    cost = np.zeros(1600)
    cost[1200:1300] = 5000.0  # everything piled up in 100 cells of the last-but-one subdomain
    cuts = np.arange(0, 1601, 200)
    new, before, after = oa.recut_plan(cuts, cost, max_shift=40, min_size=8)
    assert np.all(np.abs(new - cuts) <= 40)                       # max_shift
    assert np.all(np.diff(new) >= 8)                              # yee.rs:783-785: every subdomain keeps its minimum
    assert after <= before
    # never out of the two subdomains a cut separates, whatever max_shift allows
    new2, _, _ = oa.recut_plan(cuts, cost, max_shift=10**6, min_size=8)
    for i in range(1, 8):
        assert cuts[i - 1] + 8 <= new2[i] <= cuts[i + 1] - 8, i
    assert np.all(np.diff(new2) >= 8)
    # an edge's room: no more particles of one species through a cut than its messages hold
    most = cost.copy()
    room = np.full(7, 20000.0)
    new3, _, _ = oa.recut_plan(cuts, cost, most=most, max_shift=10**6, min_size=8, edge_room=room)
    m = np.concatenate([[0.0], np.cumsum(most)])
    for i in range(1, 8):
        lo, hi = sorted((int(cuts[i]), int(new3[i])))
        if lo != hi:  # the cells that change hands and the cell either side (a particle moves up to a cell in the hand-over step)
            assert m[min(hi + 1, 1600)] - m[max(lo - 1, 0)] <= 20000.0, i
    assert not np.array_equal(new3, cuts)  # ... but what fits does move
    # (where even one cell and its neighbours are too many, the cut stays)
    new4, _, _ = oa.recut_plan(cuts, cost, most=most, max_shift=10**6, min_size=8, edge_room=np.full(7, 12000.0))
    for i in range(1, 8):
        lo, hi = sorted((int(cuts[i]), int(new4[i])))
        assert lo == hi or m[min(hi + 1, 1600)] - m[max(lo - 1, 0)] <= 12000.0, i
    # iterating converges to a balanced partition of the pile
    k = cuts
    for _ in range(60):
        k, _, a = oa.recut_plan(k, cost, max_shift=40, min_size=8)
    assert a < 1.3, (a, k)


def test_degenerate_inputs():
    cost = np.zeros(64)
    cuts = np.array([0, 32, 64])
    new, b, a = oa.recut_plan(cuts, cost)      # no particles anywhere: nothing to balance
    assert np.array_equal(new, cuts) and b == 1.0 and a == 1.0
    one, b, a = oa.recut_plan(np.array([0, 64]), np.ones(64))
    assert np.array_equal(one, [0, 64])
    with pytest.raises(oa.OpalError):
        oa.recut_plan(np.array([0, 4, 64]), np.ones(64), min_size=8)  # a subdomain below the minimum
    # two subdomains at the minimum size cannot give anything away
    tight, _, _ = oa.recut_plan(np.array([0, 8, 16]), np.array([9.0] * 8 + [1.0] * 8), min_size=8)
    assert np.array_equal(tight, [0, 8, 16])


def test_every_subdomain_arrives_at_the_same_plan():
    """The plan is a function of the all-reduced histogram alone: whoever computes it gets the same cuts (bitwise the same
    inputs on every subdomain; no rank-dependent tie-break)."""
    rng = np.random.default_rng(11)
    cost = rng.random(3000) * (np.arange(3000) > 700)
    cuts = np.array([0, 900, 1400, 1900, 2400, 3000])
    a = [oa.recut_plan(cuts, cost, max_shift=64)[0] for _ in range(3)]
    assert all(np.array_equal(a[0], x) for x in a)
    assert loads(a[0], cost).max() < loads(cuts, cost).max()
