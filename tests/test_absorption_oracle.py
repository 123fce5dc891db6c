"""Photon absorption / stimulated emission on the CPU side: the oracle's restatement of
airy_ai_for_positive, the two scaled cross sections, Photon::interacts_with and absorb.

Pinned by the reference's own known answers for the Airy function (qed/special_functions/airy.rs:77-112, relative
1e-12: the tightest vectors the reference holds for any function on this path); the cross sections and `absorb`
have no reference test, so they are checked against an independent numpy/scipy evaluation of the same formulae and
through conservation laws.
"""
import math
import os

import numpy as np
import pytest
from scipy.special import airy

from oracle import oracle as O

ME_MEV = 0.510999
C = 2.997925e8
R_E = 2.817940e-15
ALPHA = 7.29735257e-3


# --------------------------------------------------------------------------- airy.rs:77-112
@pytest.mark.parametrize("x,target", [(0.0, 0.3550280538878172), (2.0, 0.03492413042327438),
                                      (17.0, 7.05019729838861e-22), (20.0, 1.69167286867e-27)])
def test_airy_known_answers(x, target):
    """airy_0, airy_2, airy_17, airy_20 with the reference's MAX_REL_ERR = 1e-12."""
    val = O.airy_ai(x)
    assert abs((val - target) / target) < 1.0e-12


def test_airy_200_is_none():
    """airy_200 (#[should_panic] on unwrap): the result would underflow."""
    assert O.airy_ai(200.0) is None
    assert O.airy_ai(50.0) is None and O.airy_ai(49.999) is not None
    assert O.airy_ai(-1e-9) is None


def test_airy_against_scipy_across_the_four_branches():
    xs = np.concatenate([np.linspace(0.0, 0.999999, 200), np.linspace(1.0, 1.999999, 200), np.linspace(2.0, 9.999999, 200),
                         np.linspace(10.0, 49.9, 200)])
    got = np.array([O.airy_ai(float(x)) for x in xs])
    assert np.max(np.abs(got / airy(xs)[0] - 1.0)) < 2e-12


def _load(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(os.path.basename(path)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _numbers(text):
    import re
    num = re.compile(r"-?\d\.\d+e[+-]\d+")
    body = text.split("\n", 1)[1]  # (the first line names the tool that wrote the file)
    return re.sub(num, "#", body), np.array(num.findall(body), dtype=float)


def test_committed_airy_tables_are_the_reference_numbers(tmp_path):
    """opal_amd/csrc/airy_tables.inc carries the reference's digits (tools/transcribe_airy_tables.py re-parses
    airy.rs:42,115-278): bit for bit what the reference's file parses to."""
    if not os.path.exists("/root/reference/src/qed/special_functions/airy.rs"):
        pytest.skip("the reference tree is not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = _load(os.path.join(root, "tools", "transcribe_airy_tables.py"))
    out = tmp_path / "airy_tables.inc"
    gen.main(str(out))
    assert out.read_text() == open(os.path.join(root, "opal_amd", "csrc", "airy_tables.inc")).read()
    t = gen.load()
    assert t["scale"] == 0.262183997088323 and t["series_used"] == 14 and t["series_a"][0] == 3.550280538878172e-1
    assert t["gl"][4][0][0] == pytest.approx(0.5, abs=0.3) and len(t["gl"][40][1]) == 40


def test_airy_tables_agree_with_the_computed_constants(tmp_path):
    """The mathematical cross-check: tools/make_airy_tables.py computes the same constants (scipy); the committed
    (transcribed) file has the same layout and agrees to the reference's printed precision -- 16 digits, and ~1e-11
    on weights that span 60 decades (the smallest, 1e-61, contribute nothing)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = _load(os.path.join(root, "tools", "make_airy_tables.py"))
    out = tmp_path / "airy_tables.inc"
    gen.main(str(out))
    (shape_a, a), (shape_b, b) = _numbers(out.read_text()), _numbers(open(os.path.join(root, "opal_amd", "csrc", "airy_tables.inc")).read())
    assert shape_a == shape_b
    assert np.max(np.abs(a / b - 1.0)) < 1e-11
    big = np.abs(b) > 1e-20
    assert np.max(np.abs(a[big] / b[big] - 1.0)) < 5e-14


# --------------------------------------------------------------------------- cross sections
def _sigma_numpy(k, p, cg, ce, stimulated):
    """photon_absorption.rs:16-34 / stimulated_emission.rs:16-37 written out independently with scipy's Ai."""
    if cg <= 0 or ce <= 0:
        return None
    if stimulated and (cg >= ce or k[0] >= p[0]):
        return None
    d = ce * (ce - cg) if stimulated else ce * (ce + cg)
    g = 0.5 + 0.25 * cg ** 2 / d
    z = (cg / d) ** (2.0 / 3.0)
    kp = k[0] * p[0] - k[1] * p[1] - k[2] * p[2] - k[3] * p[3]
    zbar = 2.0 * z * ce * kp / cg
    if zbar < 0 or zbar >= 50.0:
        return None
    zbar_z = 2.0 * p[0] * kp / k[0]
    return (2.0 * math.pi * R_E) ** 2 * ce * z * (4.0 * g * zbar_z - 1.0) * airy(zbar)[0] / (ALPHA * cg * k[0] * p[0])


def _collinear_pairs(n, seed):
    """Photons within ~1/gamma of an electron's direction: the only geometry where k.p is small enough for the
    Airy argument to be O(1), i.e. where these processes happen at all."""
    rng = np.random.default_rng(seed)
    gamma = 10.0 ** rng.uniform(1.0, 3.0, n)
    u = np.sqrt(gamma ** 2 - 1.0)
    theta = rng.uniform(0.0, 3.0, n) / gamma
    phi = rng.uniform(0.0, 2 * math.pi, n)
    k0 = gamma * 10.0 ** rng.uniform(-3.0, -0.05, n)
    p = np.stack([gamma, -u, np.zeros(n), np.zeros(n)], 1)
    k = np.stack([k0, -k0 * np.cos(theta), k0 * np.sin(theta) * np.cos(phi), k0 * np.sin(theta) * np.sin(phi)], 1)
    chi_e = 10.0 ** rng.uniform(-2.0, 0.5, n)
    chi_g = chi_e * k0 / gamma * rng.uniform(0.5, 1.5, n)
    return k, p, chi_g, chi_e


def test_cross_sections_against_an_independent_evaluation():
    k, p, cg, ce = _collinear_pairs(600, 3)
    seen = {"abs": 0, "st": 0, "none": 0}
    for i in range(len(cg)):
        got = O.cross_sections(k[i], p[i], float(cg[i]), float(ce[i]))
        for which, stim in ((0, False), (1, True)):
            want = _sigma_numpy(k[i], p[i], cg[i], ce[i], stim)
            if want is None:
                assert got[which] is None
                seen["none"] += 1
            else:
                assert got[which] == pytest.approx(want, rel=5e-12)
                seen["st" if stim else "abs"] += 1
    assert seen["abs"] > 200 and seen["st"] > 100 and seen["none"] > 20
    # None when either quantum parameter vanishes, or (stimulated) when the photon outweighs the electron
    assert O.cross_sections(k[0], p[0], 0.0, 0.3) == (None, None)
    assert O.cross_sections(k[0], p[0], 0.1, 0.0) == (None, None)
    kk = k[0].copy(); kk[0] = p[0][0] * 1.1
    assert O.cross_sections(kk, p[0], 0.1, 0.3)[1] is None


# --------------------------------------------------------------------------- absorb
def _populations(n_e, n_ph, nx, seed, tau_abs, tau_st, dx=1.0e-8, gamma=200.0):
    rng = np.random.default_rng(seed)
    dt = 0.95 * dx / C
    cell_e = np.sort(rng.integers(0, nx, n_e))
    ux = -math.sqrt(gamma ** 2 - 1.0) * (1.0 + 0.01 * rng.normal(size=n_e))
    e = O.Population.from_soa(O.ELECTRON, cell_e, rng.random(n_e), ux, 0.3 * rng.normal(size=n_e), 0.3 * rng.normal(size=n_e),
                              1.0e21 * (1.0 + rng.random(n_e)), dx, dt)
    e.set_electron_state(tau=np.full(n_e, 50.0), ids=np.arange(n_e, dtype=np.uint64) + 7)
    e.set_electron_chi(10.0 ** rng.uniform(-1.0, 0.0, n_e))
    cell_p = rng.integers(0, nx, n_ph)       # deliberately not sorted: the photons' order must not matter
    k0 = gamma * 10.0 ** rng.uniform(-2.0, -0.3, n_ph)
    th, phi = rng.uniform(0, 2.0, n_ph) / gamma, rng.uniform(0, 2 * math.pi, n_ph)
    ph = O.Population.from_soa(O.PHOTON, cell_p, rng.random(n_ph), -k0 * np.cos(th), k0 * np.sin(th) * np.cos(phi),
                               k0 * np.sin(th) * np.sin(phi), 1.0e19 * (1.0 + rng.random(n_ph)), dx, dt)
    ph.set_photon_state(chi=0.3 * k0 / gamma * rng.uniform(0.5, 1.0, n_ph), tau_abs=np.full(n_ph, tau_abs),
                        tau_st=np.full(n_ph, tau_st), birth_time=rng.uniform(0.0, 1e-15, n_ph),
                        ids=np.arange(n_ph, dtype=np.uint64) + 100000, y=1e-7 * rng.normal(size=n_ph), z=np.zeros(n_ph))
    return e, ph, dx, dt


def _momentum(pop, mass_mev=ME_MEV):
    s = pop.soa()
    return np.array([np.sum(s["w"] * s[c]) for c in ("ux", "uy", "uz")]) * mass_mev


def test_absorb_moves_momentum_from_photons_to_electrons():
    """interactions.rs:290-332: absorbed photons disappear and their momentum, weighted, lands on the electron;
    a stimulated emission appends a photon with the ELECTRON's weight and the same k, which the electron pays for."""
    e, ph, dx, dt = _populations(3000, 4000, 24, 5, tau_abs=4e-6, tau_st=4e-6)
    pe0, pp0 = _momentum(e), _momentum(ph)
    n_ph0 = len(ph)
    before = ph.soa()
    ids0 = ph.photon_extras()["id"]
    na, ns = O.absorb(e, ph, seed=9, step=4, t=2.0e-15, dt=dt, dx=dx)
    assert na > 100 and ns > 10 and len(ph) == n_ph0 - na + ns
    pe1, pp1 = _momentum(e), _momentum(ph)
    assert np.allclose(pe1 + pp1, pe0 + pp0, rtol=1e-12, atol=1e-12 * np.abs(pe0).max())
    ex = ph.photon_extras()
    new = slice(len(ph) - ns, len(ph))
    assert np.all(ex["birth_time"][new] == 2.0e-15)            # .at_time(t)
    # every stimulated photon has an identity of its own (keyed by the photon that stimulated it), so that siblings off
    # one electron, and the photon that electron may radiate spontaneously in the same step, draw independently later
    assert len(np.unique(ex["id"])) == len(ex["id"]) and np.all(ex["id"][new] > 100000)
    assert np.all(np.isin(ph.soa()["w"][new], e.soa()["w"]))    # with_weight(e.weight()): an electron's weight, not a photon's
    assert not np.any(np.isin(ph.soa()["w"][new], before["w"]))
    ta, ts = ph.photon_depths()
    assert np.all(ta[new] > 0) and np.all(ts[new] > 0)          # with_optical_depths
    # the survivors keep their k; their optical depths only went down (or were reset by a stimulated emission)
    surv = ph.soa()
    assert set(np.round(surv["ux"][:len(ph) - ns], 9)).issubset(set(np.round(before["ux"], 9)))
    assert np.array_equal(np.sort(ex["id"][:len(ph) - ns]), np.sort(np.setdiff1d(ids0, np.setdiff1d(ids0, ex["id"][:len(ph) - ns]))))
    # gamma follows the kick (electron.rs:261)
    s = e.soa()
    assert np.allclose(s["aux"], np.sqrt(1 + s["ux"] ** 2 + s["uy"] ** 2 + s["uz"] ** 2), rtol=1e-15)


def test_absorb_filters_and_empty_cells():
    e, ph, dx, dt = _populations(500, 2000, 16, 8, tau_abs=1e-6, tau_st=50.0)
    # stop_time: photons older than the limit are left alone (interactions.rs:188-192)
    births = ph.photon_extras()["birth_time"]
    cut = float(np.median(births))
    t = 1.5e-15
    e1, ph1, _, _ = _populations(500, 2000, 16, 8, tau_abs=1e-6, tau_st=50.0)
    na_all, _ = O.absorb(e, ph, 1, 0, t, dt, dx)
    na_young, _ = O.absorb(e1, ph1, 1, 0, t, dt, dx, stop_time=t - cut)
    assert 0 < na_young < na_all
    left = ph1.photon_extras()["birth_time"]
    assert np.sum(births < cut) == np.sum(left < cut)           # none of the old ones was touched
    # max_displacement on hypot(y, z) (:195-199)
    e2, ph2, _, _ = _populations(500, 2000, 16, 8, tau_abs=1e-6, tau_st=50.0)
    na_near, _ = O.absorb(e2, ph2, 1, 0, t, dt, dx, max_displacement=0.5e-7)
    assert 0 < na_near < na_all
    far_before = np.sum(np.abs(_populations(500, 2000, 16, 8, 1e-6, 50.0)[1].photon_extras()["y"]) > 0.5e-7)
    assert np.sum(np.abs(ph2.photon_extras()["y"]) > 0.5e-7) == far_before > 0   # none of the far ones was touched
    # photons with chi / energy below the cut-off never interact (:184), nor do photons in cells without electrons
    e3, ph3, _, _ = _populations(500, 2000, 16, 8, tau_abs=1e-6, tau_st=50.0)
    ph3.set_photon_state(chi=np.full(len(ph3), 1e-12))
    assert O.absorb(e3, ph3, 1, 0, t, dt, dx) == (0, 0)
    e4 = O.Population.empty(O.ELECTRON)
    assert O.absorb(e4, ph3, 1, 0, t, dt, dx) == (0, 0)


def test_optical_depths_run_down_by_the_summed_probabilities():
    """photon.rs:319-339: without an event, tau_abs and tau_st drop by w_e (c dt/dx) sigma summed over the
    electrons of the photon's cell."""
    e, ph, dx, dt = _populations(400, 300, 8, 12, tau_abs=1.0, tau_st=1.0)
    es, ps, pe = e.soa(), ph.soa(), ph.photon_extras()
    chi_e = e.electron_extras()["chi"]
    assert O.absorb(e, ph, 2, 1, 0.0, dt, dx) == (0, 0)
    ta, ts = ph.photon_depths()
    for i in (0, 17, 123, 299):
        k0 = math.sqrt(ps["ux"][i] ** 2 + ps["uy"][i] ** 2 + ps["uz"][i] ** 2)
        k = [k0, ps["ux"][i], ps["uy"][i], ps["uz"][i]]
        sa = ss = 0.0
        for j in np.flatnonzero(es["cell"] == ps["cell"][i]):
            p = [es["aux"][j], es["ux"][j], es["uy"][j], es["uz"][j]]
            a, b = O.cross_sections(k, p, float(pe["chi"][i]), float(chi_e[j]))
            sa += es["w"][j] * (C * dt / dx) * (a or 0.0)
            ss += es["w"][j] * (C * dt / dx) * (b or 0.0)
        assert sa > 0 and ss >= 0
        assert 1.0 - ta[i] == pytest.approx(sa, rel=1e-9)   # (1 - x keeps ~11 digits of a 1e-5 sum)
        assert 1.0 - ts[i] == pytest.approx(ss, rel=1e-9, abs=1e-18)


def test_absorption_randoms_are_keyed_by_photon_and_step():
    a = O.absorption_randoms(1, 5, 1e-15, 3)
    assert np.all((a >= 0) & (a < 1))
    assert not np.array_equal(a, O.absorption_randoms(1, 5, 2e-15, 3))   # a sibling photon of the same parent
    assert not np.array_equal(a, O.absorption_randoms(1, 6, 1e-15, 3))
    assert not np.array_equal(a, O.absorption_randoms(1, 5, 1e-15, 4))
    assert np.array_equal(a, O.absorption_randoms(1, 5, 1e-15, 3))
    assert not np.array_equal(O.birth_randoms(1, 5, 3), O.emission_randoms(1, 5, 3)[:2])
