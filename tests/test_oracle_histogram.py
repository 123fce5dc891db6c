"""The oracle's histogram (the checker of tests/test_gpu_histograms.py) against numpy on hand-placed values: the
reference's own histogram tests are #[ignore]d (they need an MPI launch), so what they assert -- one point gives a
histogram (hgram.rs:437-487) -- and the binning rules hgram.rs:131-143, 168-262 state are checked here without the
reference.  (An empty population gives None: the wrapper's caller sees n = 0 bins.)"""
import numpy as np

C = 2.997925e8
MEV = 0.510999  # px = u * 0.510999 MeV/c


def _pop(o, ux, w, nx=16, dx=1.0e-8):
    dt = 0.95 * dx / C
    n = len(ux)
    cell = np.sort(np.arange(n) % nx).astype(np.int64)
    z = np.zeros(n)
    return o.Grid.single(nx, 0.0, dx), o.Population.from_soa(o.ELECTRON, cell, np.full(n, 0.5), ux, z, z, w, dx, dt)


def test_one_point_gives_one_bin_of_volume_one(oracle):
    o = oracle
    g, p = _pop(o, np.array([0.3]), np.array([7.0]))
    h = o.histogram(p, g, ["px"])
    assert h["nbins"][0] == 1 and len(h["cts"]) == 1          # hgram.rs:131-134: min == max
    assert h["cts"][0] == 7.0 and h["total"] == 7.0           # density with bin volume 1 (hgram.rs:202-204)
    assert h["bin_sz"][0] == 0.0                              # hgram.rs:253
    h2 = o.histogram(p, g, ["x", "px"])
    assert h2["nbins"] == [1, 1] and h2["cts"][0] == 7.0


def test_fixed_and_automatic_bins_against_numpy(oracle):
    o = oracle
    rng = np.random.default_rng(3)
    n = 5000
    lo, hi, size = -2.4, 2.45, 0.25                           # range 4.85: ceil(19.4) = 20 bins of the fixed size
    v = lo + (rng.integers(0, 19, n) + rng.uniform(0.05, 0.95, n)) * size
    v[0], v[1] = lo, hi
    ux = v / MEV
    v = ux * MEV
    w = 1.0 + rng.random(n)
    g, p = _pop(o, ux, w)
    h = o.histogram(p, g, ["px"], 0.25)
    assert h["nbins"][0] == 20 and h["bin_sz"][0] == size
    want = np.bincount(np.floor((v - v.min()) / size).astype(int), weights=w, minlength=20) / size
    assert np.max(np.abs(h["cts"] - want)) <= 1e-12 * np.max(want)
    assert abs(h["total"] - np.sum(w)) <= 1e-12 * np.sum(w)
    for bspec, nb in ((17, 17), ("auto", int(np.ceil(2.0 * n ** (1.0 / 3.0))))):
        h = o.histogram(p, g, ["px"], bspec)
        assert h["nbins"][0] == nb
        bs = (v.max() - v.min()) / nb
        assert abs(h["bin_sz"][0] / bs - 1.0) < 1e-14
        # the maximum sits on the last edge and is dropped by `fbin < len` (hgram.rs:236) but counted in the total
        assert abs(np.sum(h["cts"]) * bs - (np.sum(w) - w[1])) <= 1e-9 * np.sum(w)
        assert abs(h["total"] - np.sum(w)) <= 1e-12 * np.sum(w)
