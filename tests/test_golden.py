"""Regression freeze of the oracle (tests/golden/pic_step_vectors.json is written BY the oracle, see make_golden.py):
the oracle must still reproduce it bit for bit, i.e. it has not drifted; the HIP path must match it to 1e-12.
This is not the parity pin -- that is tests/test_golden_independent.py, whose vectors come from a second,
independent restatement of the reference."""
import json
import os

import numpy as np
import pytest

C = 2.997925e8
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "pic_step_vectors.json")))


def test_oracle_weight_flux(oracle, gold):
    for a, w in gold["weight"]:
        assert oracle.weight(a) == w
    for xi, xf, f in gold["flux"]:
        assert oracle.flux(xi, xf) == f
    # properties the reference claims (yee.rs:131-132, 179-184): weights sum to 1, flux conserves weight
    for x in np.linspace(0.0, 0.999, 37):
        s3 = oracle.weight(0.5 + x) + oracle.weight(0.5 - x) + oracle.weight(1.5 - x)
        s4 = oracle.weight(1.0 + x) + oracle.weight(x) + oracle.weight(1.0 - x) + oracle.weight(2.0 - x)
        assert abs(s3 - 1.0) < 1e-15 and abs(s4 - 1.0) < 1e-15
        for d in (-0.9, -0.3, 0.2, 0.7):
            xn = x + d
            # continuity (yee.rs:179-184): the change of the weight at the edge midway between two
            # boundaries equals flux in minus flux out
            for k in (-1, 0, 1, 2):
                dW = oracle.weight(k - xn) - oracle.weight(k - x)
                net = oracle.flux(k - 0.5 - x, k - 0.5 - xn) - oracle.flux(k + 0.5 - x, k + 0.5 - xn)
                assert abs(dW - net) < 1e-14
            tot = sum(oracle.flux(k + 0.5 - x, k + 0.5 - xn) for k in range(-3, 5))
            assert abs(tot - d) < 1e-14  # total flux through all boundaries = displacement


def test_oracle_pushes(oracle, gold):
    o = oracle
    for v in gold["electron_push"]:
        e = o.electron_create(5, v["x"], v["u"], 1.0, v["dx"], v["dt"])
        o.electron_push(e, v["E"], v["B"], v["dx"], v["dt"])
        w = v["out"]
        assert (e.cell, e.x, e.prev_x, list(e.u), e.gamma) == (w["cell"], w["x"], w["prev_x"], w["u"], w["gamma"])
        assert (e.chi, e.work, e.y, e.z) == (w["chi"], w["work"], w["y"], w["z"])
    for v in gold["ion_push"]:
        p = o.ion_create(5, v["x"], v["u"], 1.0, v["dx"], v["dt"])
        p.Z_star, p.Z, p.A = v["Z"], v["Z"], v["A"]
        o.ion_push(p, v["E"], v["B"], v["dx"], v["dt"])
        w = v["out"]
        assert (p.cell, p.x, p.prev_x, list(p.u), p.gamma_m1) == (w["cell"], w["x"], w["prev_x"], w["u"], w["gamma_m1"])


def test_oracle_deposit_and_field_step(oracle, gold):
    o = oracle
    for v in gold["deposit"]:
        g = o.Grid.single(v["nx"], 0.0, v["dx"])
        p = o.Population.from_soa(o.ELECTRON, [8], [v["x0"]], [v["u"][0]], [v["u"][1]], [v["u"][2]], [v["w"]], v["dx"], v["dt"])
        p.advance(g, v["dt"])
        g.clear()
        g.deposit(p, v["dt"])
        for c in ("jx", "jy", "jz", "rho"):
            assert g.raw(c).tolist() == v[c], c
        # the reference's quirks: nothing lands at index+2 for jy/jz; rho's 4th tap sits at index-2
        idx = v["cell"] + 4
        assert v["jy"][idx + 2] == 0.0 and v["jz"][idx + 2] == 0.0
        if v["x"] > 0.5:
            assert v["rho"][idx - 2] != 0.0 and v["rho"][idx + 2] == 0.0
    fs = gold["field_step"]
    g = o.Grid.single(fs["nx"], 0.0, fs["dx"])
    for c, a in fs["init"].items():
        g.set_raw(c, np.array(a))
    g.synchronize(None, None, 0.0, fs["dt"])
    g.advance(fs["dt"])
    for c, a in fs["out"].items():
        assert g.raw(c).tolist() == a, c
    assert g.em_field_energy() == fs["energy"]


@pytest.mark.gpu
def test_hip_against_golden(gold):
    import opal_amd as oa
    # deposit stencils + the push that precedes them
    for v in gold["deposit"]:
        g = oa.Grid.single(v["nx"], 0.0, v["dx"])
        p = oa.Population.from_soa(oa.ELECTRON, [8], [v["x0"]], [v["u"][0]], [v["u"][1]], [v["u"][2]], [v["w"]], v["dx"], v["dt"])
        p.set_sort_interval(0)
        p.advance(g, v["dt"])
        s = p.soa(prev_x=True)
        assert int(s["cell"][0]) == v["cell"]
        assert abs(s["x"][0] - v["x"]) < 1e-13 and abs(s["prev_x"][0] - v["prev_x"]) < 1e-13
        g.clear()
        g.deposit(p, v["dt"])
        for c in ("jx", "jy", "jz", "rho"):
            ref = np.array(v[c])
            scale = np.max(np.abs(ref)) if np.max(np.abs(ref)) > 0 else 1.0
            assert np.max(np.abs(g.raw(c) - ref)) <= 1e-12 * scale, c
            assert np.array_equal(g.raw(c) == 0.0, ref == 0.0), c  # same stencil footprint
        # the fused pass gives the same stencil
        g2 = oa.Grid.single(v["nx"], 0.0, v["dx"])
        p2 = oa.Population.from_soa(oa.ELECTRON, [8], [v["x0"]], [v["u"][0]], [v["u"][1]], [v["u"][2]], [v["w"]], v["dx"], v["dt"])
        p2.set_sort_interval(0)
        p2.advance_deposit(g2, v["dt"])
        for c in ("jx", "jy", "jz", "rho"):
            ref = np.array(v[c])
            scale = np.max(np.abs(ref)) if np.max(np.abs(ref)) > 0 else 1.0
            assert np.max(np.abs(g2.raw(c) - ref)) <= 1e-12 * scale, c
    # single-particle pushes in uniform fields
    for species, key in ((oa.ELECTRON, "electron_push"), (oa.ION, "ion_push")):
        for v in gold[key]:
            g = oa.Grid.single(16, 0.0, v["dx"])
            for k, c in enumerate(("Ex", "Ey", "Ez")):
                g.set_raw(c, np.full(g.len, v["E"][k]))
            for k, c in enumerate(("Bx", "By", "Bz")):
                g.set_raw(c, np.full(g.len, v["B"][k]))
            p = oa.Population.from_soa(species, [5], [v["x"]], [v["u"][0]], [v["u"][1]], [v["u"][2]], [1.0], v["dx"], v["dt"])
            p.set_sort_interval(0)
            if species == oa.ION:
                p.set_charge_state(v["Z"], v["Z"], v["A"])
            else:
                p.track_extras(True)
            p.advance(g, v["dt"])
            s = p.soa(prev_x=True)
            w = v["out"]
            assert int(s["cell"][0]) == w["cell"]
            assert abs(s["x"][0] - w["x"]) < 1e-12 and abs(s["prev_x"][0] - w["prev_x"]) < 1e-12
            un = np.array([s["ux"][0], s["uy"][0], s["uz"][0]])
            assert np.max(np.abs(un - np.array(w["u"]))) <= 1e-12 * np.max(np.abs(w["u"]))
            if species == oa.ELECTRON:
                ex = p.extras()
                assert abs(ex["chi"][0] - w["chi"]) <= 1e-10 * abs(w["chi"])
                assert abs(ex["work"][0] - w["work"]) <= 1e-12 * abs(w["work"])
                assert abs(ex["y"][0] - w["y"]) <= 1e-12 * abs(w["y"]) and abs(ex["z"][0] - w["z"]) <= 1e-12 * abs(w["z"])
    # one synchronize + advance of a random periodic grid
    fs = gold["field_step"]
    g = oa.Grid.single(fs["nx"], 0.0, fs["dx"])
    for c, a in fs["init"].items():
        g.set_raw(c, np.array(a))
    g.synchronize(None, None, 0.0, fs["dt"])
    g.advance(fs["dt"])
    for c, a in fs["out"].items():
        ref = np.array(a)
        assert np.max(np.abs(g.raw(c) - ref)) <= 1e-13 * np.max(np.abs(ref)), c
    assert abs(g.em_field_energy() - fs["energy"]) <= 1e-13 * fs["energy"]


# ----------------------------------------------------------------------------- photon emission
@pytest.fixture(scope="module")
def qgold():
    return json.load(open(os.path.join(HERE, "golden", "qed_emission_vectors.json")))


def test_oracle_emission_vectors(oracle, qgold):
    """The oracle (and the generated sampling tables it reads) against the committed emission vectors.
    Integer work bit for bit; floating point to 1e-12 (libm's log / exp / pow / acos may differ in the last
    digit between glibc versions)."""
    o = oracle
    for v in qgold["philox"]:
        assert list(o.philox(v["ctr"], v["key"])) == v["out"]
    for v in qgold["randoms"]:
        assert list(o.emission_randoms(v["seed"], v["id"], v["step"])) == v["out"]
    for v in qgold["rate"]:
        assert o.rate(v["chi"], v["gamma"]) == pytest.approx(v["out"], rel=1e-13)
    for v in qgold["sample"]:
        got = o.qed_sample(v["chi"], v["gamma"], *v["r"])
        assert got[0] == pytest.approx(v["out"][0], rel=1e-12) and got[2] == v["out"][2]
        assert got[1] * v["gamma"] == pytest.approx(v["out"][1] * v["gamma"], rel=1e-9, abs=1e-12)
    for v in qgold["classical"]:
        got = o.qed_classical_sample(v["chi"], v["gamma"], *v["r"])
        assert got[0] == pytest.approx(v["out"][0], rel=1e-12)
        assert got[1] * v["gamma"] == pytest.approx(v["out"][1] * v["gamma"], rel=1e-9, abs=1e-12)


@pytest.mark.gpu
def test_hip_emission_vectors(qgold):
    import opal_amd as oa
    for v in qgold["randoms"]:
        assert list(oa.qed_randoms(v["seed"], [v["id"]], v["step"])[0]) == v["out"]
    r = qgold["rate"]
    got = oa.qed_rate([v["chi"] for v in r], np.array([v["gamma"] for v in r]))
    assert np.max(np.abs(got / np.array([v["out"] for v in r]) - 1.0)) < 1e-13
    for key, classical in (("sample", False), ("classical", True)):
        s = qgold[key]
        rr = np.array([v["r"] for v in s])
        gam = np.array([v["gamma"] for v in s])
        got = oa.qed_sample(np.array([v["chi"] for v in s]), gam, rr[:, 0], rr[:, 1], rr[:, 2], classical=classical)
        want = np.array([v["out"] for v in s])
        assert np.max(np.abs(got[:, 0] / want[:, 0] - 1.0)) < 1e-7
        assert np.max(np.abs(got[:, 1] - want[:, 1]) * gam / np.maximum(want[:, 1] * gam, 1e-3)) < 1e-6
        assert np.array_equal(got[:, 2], want[:, 2])
