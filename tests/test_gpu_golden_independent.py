"""The HIP operators, through the C ABI, against tests/golden/pic_step_vectors_independent.json — vectors written by a
pure-Python restatement of the reference that shares nothing with oracle/ or the kernels
(tests/golden/make_golden_pic_independent.py; the oracle's own leg is tests/test_golden_independent.py, bit for bit).

No oracle in this file: every expected value is read from the committed fixture.

Tolerance: 1e-12 of each quantity's scale for one operator from the given state (DESIGN.md §2 lists the six ways the
kernels' arithmetic differs from the reference's order: contraction, reciprocals, recomputed gamma, summation order,
branch-free weights, closed-form flux); integers (cells, counts, who migrates where, grid designs) exact.  The
multi-step worlds (up to 8 steps, fields of 2e11 V/m acting on |u| ~ 1 particles) are held to 1e-10: rounding
differences of one step are amplified by the following ones.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu

TOL = 1e-12
TOL_RUN = 1e-10
FIELDS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz")
SOURCES = ("rho", "jx", "jy", "jz")
BDY = {"internal": 0, "laser": 1, "absorbing": 2, "conducting": 3}


def fh(v):
    if isinstance(v, str):
        return float.fromhex(v)
    if isinstance(v, (float, np.ndarray)):
        return v
    return np.array([fh(a) for a in v], dtype=np.float64)


def folded(dump, size):
    """What a lone periodic subdomain's synchronize makes of deposited sources (yee.rs:353-354, 365-370, 424-452: the
    first and last 8 cells are each other's messages; rho and j are added on all 8), in numpy on the fixture's arrays."""
    out = {}
    for c, a in dump.items():
        a = fh(a).copy()
        pre, n = a.copy(), len(a)
        assert n == size + 8
        a[0:4] += pre[n - 8:n - 4]
        a[4:8] += pre[n - 4:n]
        a[n - 8:n - 4] += pre[0:4]
        a[n - 4:n] += pre[4:8]
        out[c] = a
    return out


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "pic_step_vectors_independent.json")))


@pytest.fixture(scope="module")
def oa():
    import opal_amd
    assert opal_amd.device_count() >= 1
    return opal_amd


def build(oa, design, r, stream=None):
    return oa.Grid(r, design["numtasks"], BDY[design["left"]], BDY[design["right"]], fh(design["xmin"]),
                   fh(design["dx"]), np.array(design["nx"], dtype=np.uint64), fh(design["offset"]), stream)


def load(g, dump):
    for c, a in dump.items():
        g.set_raw(c, fh(a))


def close(got, want, scale, tol, what):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = float(np.max(np.abs(got - want))) if got.size else 0.0
    assert err <= tol * scale, (what, err, scale, err / scale if scale else None)


def group_scale(dump, comps):
    return max(max(float(np.max(np.abs(fh(dump[c])))) for c in comps), 1e-300)


def check_grid(g, want, tol, what, groups=(("Ex", "Ey", "Ez"), ("Bx", "By", "Bz"), ("jx", "jy", "jz"), ("rho",))):
    """Each field group against its own scale (E in V/m, B in T, j, rho)."""
    for comps in groups:
        comps = [c for c in comps if c in want]
        if not comps:
            continue
        s = group_scale(want, comps)
        for c in comps:
            close(g.raw(c), fh(want[c]), s, tol, f"{what}: {c}")


def test_fields_at(oa, gold):
    """opal_grid_fields_at against yee.rs:499-529 as restated independently (stagger, spline order, Bx un-interpolated)"""
    for rec in gold["fields_at"]:
        g = build(oa, rec["design"], rec["id"])
        load(g, rec["fields"])
        q = rec["queries"]
        E, B = g.fields_at([v["c"] for v in q], [fh(v["xi"]) for v in q])
        se, sb = group_scale(rec["fields"], ("Ex", "Ey", "Ez")), group_scale(rec["fields"], ("Bx", "By", "Bz"))
        close(E, np.array([fh(v["E"]) for v in q]), se, TOL, f"{rec['name']}: E")
        close(B, np.array([fh(v["B"]) for v in q]), sb, TOL, f"{rec['name']}: B")


def _uniform_grid(oa, E, B, dx):
    g = oa.Grid.single(32, 0.0, dx)
    for k, c in enumerate(("Ex", "Ey", "Ez")):
        g.set_raw(c, np.full(g.len, E[k]))
    for k, c in enumerate(("Bx", "By", "Bz")):
        g.set_raw(c, np.full(g.len, B[k]))
    return g


@pytest.mark.parametrize("species", ["electron", "ion"])
def test_pushes(oa, gold, species):
    """opal_pop_advance on one particle in a uniform field (the gather of a uniform field is that field to an ulp):
    Vay for electrons with chi, work, y, z (electron.rs:268-330), Boris for ions of four charge states
    (ion.rs:168-214), kinetic energies (electron.rs:122-126, ion.rs:124-130); three pushes each."""
    for n, v in enumerate(gold[f"{species}_push"]):
        dx, dt = fh(v["dx"]), fh(v["dt"])
        E, B = fh(v["E"]), fh(v["B"])
        u = fh(v["u"])
        g = _uniform_grid(oa, E, B, dx)
        sp = oa.ELECTRON if species == "electron" else oa.ION
        p = oa.Population.from_soa(sp, [16], [fh(v["x"])], [u[0]], [u[1]], [u[2]], [fh(v["w"])], dx, dt)
        if species == "ion":
            p.set_charge_state(fh(v["Z_star"]), fh(v["Z"]), fh(v["A"]))
        else:
            p.track_extras(True)
        ke0 = fh(v["created"]["kinetic_energy"])
        close(p.total_kinetic_energy(), ke0, max(abs(ke0), 1e-300), TOL, f"{species} {n}: KE at creation")
        for k, w in enumerate(v["pushes"]):
            p.advance(g, dt)
            s = p.soa(prev_x=True)
            what = f"{species} {n} ({v['kind']}) push {k}"
            assert int(s["cell"][0]) - 16 == w["cell"] - 5, what
            uw = fh(w["u"])
            us = max(float(np.max(np.abs(uw))), 1e-3)
            close([s["ux"][0], s["uy"][0], s["uz"][0]], uw, us, TOL, f"{what}: u")
            close(s["x"][0], fh(w["x"]), 1.0, TOL, f"{what}: x")
            close(s["prev_x"][0], fh(w["prev_x"]), 1.0, TOL, f"{what}: prev_x")
            ke = fh(w["kinetic_energy"])
            close(p.total_kinetic_energy(), ke, max(abs(ke), abs(ke0), 1e-300), 4 * TOL, f"{what}: KE")
            if species == "electron":
                ex = p.extras()
                # chi = sqrt(|gamma E + c u x B|^2 - (E.u)^2) / E_crit (electron.rs:281): a difference of squares, so it is
                # chi^2 that is conditioned like its terms, S^2
                chi = fh(w["chi"])
                umax = max(float(np.linalg.norm(fh(q["u"]))) for q in [v["created"]] + v["pushes"]) + 1e-3
                S = (np.sqrt(1.0 + umax ** 2) * np.linalg.norm(E) + 2.997925e8 * umax * np.linalg.norm(B)) / 1.323285e18
                close(ex["chi"][0] ** 2, chi ** 2, max(S * S, 1e-300), 8 * TOL, f"{what}: chi^2")
                wk = fh(w["work"])
                close(ex["work"][0], wk, max(abs(wk), abs(fh(v["pushes"][-1]["work"])), 1e-300), 4 * TOL, f"{what}: work")
                ys = max(abs(fh(v["pushes"][-1]["y"])), abs(fh(v["pushes"][-1]["z"])), 1e-300)
                close([ex["y"][0], ex["z"][0]], [fh(w["y"]), fh(w["z"])], ys, 4 * TOL, f"{what}: y, z")


def make_pop(oa, species, created, dx, dt, charge_state=None, stream=None, shift=0):
    cell = np.array(created["cell"], dtype=np.int32) + shift
    p = oa.Population.from_soa(species, cell, fh(created["x"]), fh(created["ux"]), fh(created["uy"]),
                               fh(created["uz"]), fh(created["w"]), dx, dt, capacity=len(cell) + 4096, stream=stream)
    if charge_state is not None:
        p.set_charge_state(*charge_state)
    return p


def canon(cell, x, *rest):
    """order by (cell, x): the reference leaves the order inside a cell to an unstable sort (mod.rs:288)"""
    o = np.lexsort((x, cell))
    return [np.asarray(a)[o] for a in (cell, x) + rest]


def check_pop(p, want, tol, what, extras=False):
    s = p.soa(prev_x=True)
    wc = np.array(want["cell"], dtype=np.int64)
    assert len(s["cell"]) == len(wc), (what, len(s["cell"]), len(wc))
    got = canon(s["cell"].astype(np.int64), s["x"], s["prev_x"], s["ux"], s["uy"], s["uz"], s["w"])
    exp = canon(wc, *[fh(want[k]) for k in ("x", "prev_x", "ux", "uy", "uz", "w")])
    assert np.array_equal(got[0], exp[0]), f"{what}: cells"
    us = max(float(np.max(np.abs(np.concatenate(exp[3:6])))) if len(wc) else 1.0, 1e-3)
    for k, name in ((1, "x"), (2, "prev_x")):
        close(got[k], exp[k], 1.0, tol, f"{what}: {name}")
    for k, name in ((3, "ux"), (4, "uy"), (5, "uz")):
        close(got[k], exp[k], us, tol, f"{what}: {name}")
    assert np.array_equal(got[6], exp[6]), f"{what}: weights travel unchanged"


def fold(g, dt):
    """A single periodic subdomain's deposit may use either image of an edge cell (ghost or far end); they agree after
    synchronize has folded the ghosts (yee.rs:424-452)"""
    g.synchronize(None, None, 0.0, dt)


def test_deposit(oa, gold):
    """opal_grid_deposit: straight after upload (create()'s prev_x, electron.rs:88 / ion.rs:87) and after
    opal_pop_advance; the fused opal_pop_advance_deposit must give the second answer too."""
    for rec in gold["deposit"]:
        d = rec["design"]
        dx, dt = fh(d["dx"]), fh(rec["dt"])
        cs = fh(rec["charge_state"])
        for fused in (False, True):
            g = build(oa, d, 0)
            load(g, rec["fields"])
            el = make_pop(oa, oa.ELECTRON, rec["electrons_created"], dx, dt)
            io = make_pop(oa, oa.ION, rec["ions_created"], dx, dt, cs)
            for p in (el, io):
                p.set_sort_interval(0)
            if not fused:
                g.clear()
                g.deposit(el, dt)
                g.deposit(io, dt)
                want = folded(rec["after_create"], g.size)
                fold(g, dt)
                check_grid(g, want, TOL, f"deposit {rec['name']} after create", (("jx", "jy", "jz"), ("rho",)))
                el.advance(g, dt)
                io.advance(g, dt)
                check_pop(el, rec["electrons_advanced"], TOL, f"deposit {rec['name']}: electrons")
                check_pop(io, rec["ions_advanced"], TOL, f"deposit {rec['name']}: ions")
                g.clear()
                g.deposit(el, dt)
                g.deposit(io, dt)
            else:
                g.clear()
                el.advance_deposit(g, dt)
                io.advance_deposit(g, dt)
            want = folded(rec["after_advance"], g.size)
            fold(g, dt)
            check_grid(g, want, TOL, f"deposit {rec['name']} after advance, fused={fused}", (("jx", "jy", "jz"), ("rho",)))
            ke = rec["kinetic_energy_sequential_sum"]
            for p, k in ((el, "electrons"), (io, "ions")):
                w = fh(ke[k])
                close(p.total_kinetic_energy(), w, max(abs(w), 1e-300), 1e-11, f"{k} KE total")


class LaserProbe:
    def __init__(self, calls, value_y=None, value_z=None):
        self.want = {"y": [c for c in calls if c[0] == "y"], "z": [c for c in calls if c[0] == "z"]}
        self.value = {"y": value_y, "z": value_z}
        self.n = {"y": 0, "z": 0}
        self.errors = []

    def fn(self, comp):
        def f(t, x):
            k = self.n[comp]
            self.n[comp] += 1
            if k >= len(self.want[comp]):
                self.errors.append(f"laser_{comp} called more often than the reference would")
                return 0.0
            rec = self.want[comp][k]
            # the time and the place the laser is asked for (yee.rs:456-462, main.rs:260-266): pre-increment t, cell 2
            if abs(t - fh(rec[1])) > 1e-12 * abs(fh(rec[1])) + 1e-30 or abs(x - fh(rec[2])) > 1e-12 * abs(fh(rec[2])) + 1e-20:
                self.errors.append(f"laser_{comp} call {k}: asked at ({t!r}, {x!r}), the reference asks at "
                                   f"({fh(rec[1])!r}, {fh(rec[2])!r})")
            return fh(rec[3]) if len(rec) > 3 else self.value[comp]
        return f

    def done(self):
        assert not self.errors, self.errors[:3]
        assert self.n == {k: len(v) for k, v in self.want.items()}, (self.n, {k: len(v) for k, v in self.want.items()})


def test_synchronize_advance_energy_interpolate(oa, gold):
    """opal_grid_synchronize (+ links between subdomains), opal_grid_advance, opal_grid_em_field_energy,
    opal_grid_interpolate on worlds of random cells: every boundary kind, equal and unequal subdomains, the 8-cell
    minimum (yee.rs:351-496, 345-349, 839-867, 787-835)."""
    from _inproc_world import InProcLink, run_world
    for rec in gold["synchronize_advance"]:
        d = rec["design"]
        nt = d["numtasks"]
        t, dt = fh(rec["t"]), fh(rec["dt"])
        probe = LaserProbe(rec["laser_calls"], fh(rec["laser_y_value"]), fh(rec["laser_z_value"]))
        streams = [oa.stream_create() for _ in range(nt)]

        def body(r, w):
            st = streams[r]
            link = InProcLink(w, r, particle_capacity=64, stream=st, periodic=d["left"] == "internal") if nt > 1 else None
            g = build(oa, d, r, st)
            load(g, rec["before"][r])
            assert np.array_equal(g.raw("x"), fh(rec["cell_x"][r])) or \
                np.max(np.abs(g.raw("x") - fh(rec["cell_x"][r]))) <= 1e-15 * np.max(np.abs(fh(rec["cell_x"][r])))
            g.synchronize(probe.fn("y"), probe.fn("z"), t, dt, link)
            a = {c: g.raw(c) for c in rec["before"][r]}
            g.advance(dt)
            b = {c: g.raw(c) for c in rec["before"][r]}
            return a, b, g.em_field_energy(), g.interpolate()

        try:
            res = run_world(nt, body)
        finally:
            for st in streams:
                oa.stream_destroy(st)
        probe.done()
        for r, (a, b, energy, rows) in enumerate(res):
            what = f"{rec['name']} rank {r}"
            for got, want, stage in ((a, rec["after_synchronize"][r], "synchronize"), (b, rec["after_advance"][r], "advance")):
                for comps in (("Ex", "Ey", "Ez"), ("Bx", "By", "Bz"), ("jx", "jy", "jz"), ("rho",)):
                    # scale: the group's size before the operator (an absorber may have damped what is left of it)
                    s = group_scale(rec["before"][r], comps)
                    for c in comps:
                        if stage == "advance" and c in ("jx", "jy", "jz", "rho"):
                            continue  # the field solve leaves the sources to the next clear (the HIP path zeroes a spare block)
                        close(got[c], fh(want[c]), s, TOL, f"{what} after {stage}: {c}")
            w = fh(rec["em_field_energy_local"][r])
            close(energy, w, abs(w), TOL, f"{what}: energy")
            want_rows = np.array([fh(row) for row in rec["interpolate"][r]])
            for col, comps in ((slice(0, 1), None), (slice(5, 8), ("Ex", "Ey", "Ez")), (slice(8, 11), ("Bx", "By", "Bz"))):
                s = np.max(np.abs(want_rows[:, col]))
                close(rows[:, col], want_rows[:, col], s, TOL, f"{what}: interpolate columns {col}")


DENSITIES = {
    "uniform": lambda p: (lambda x: p[0]),
    "slab": lambda p: (lambda x: p[0] if (x >= p[1] and x < p[2]) else 0.0),
    "ramp": lambda p: (lambda x: p[0] * (x - p[1]) / (p[2] - p[1]) if (x >= p[1] and x < p[2]) else 0.0),
    "two_slabs": lambda p: (lambda x: p[0] if (x >= p[1] and x < p[2]) else (p[3] if (x >= p[4] and x < p[5]) else 0.0)),
}


def test_grid_designs(oa, gold):
    """opal_design_unbalanced / opal_design_balanced (host C++): cuts exact, offsets to the bit"""
    for v in gold["design_unbalanced"]:
        nx, off = oa.design_unbalanced(v["numtasks"], v["size"], fh(v["dx"]), v["min_subsize"])
        assert nx.tolist() == v["nx"] and np.array_equal(off, fh(v["offset"]))
    for v in gold["design_balanced"]:
        ne = DENSITIES[v["density"]]([float(a) for a in fh(v["params"])])
        nx, off = oa.design_balanced(v["numtasks"], v["size"], fh(v["xmin"]), fh(v["dx"]), ne, v["min_subsize"])
        assert nx.tolist() == v["nx"], (v["density"], v["numtasks"], nx.tolist(), v["nx"])
        assert np.array_equal(off, fh(v["offset"]))


@pytest.mark.parametrize("fused", [False, True])
def test_world_runs(oa, gold, fused):
    """main.rs:238-267 for whole small worlds, several steps: the reference's operator order and the fused pass, one
    subdomain per thread and stream, messages through the library's link."""
    from _inproc_world import InProcLink, run_world
    for rec in gold["world_runs"]:
        d = rec["design"]
        nt = d["numtasks"]
        dx, dt = fh(d["dx"]), fh(rec["dt"])
        cs = fh(rec["charge_state"])
        dep = rec["current_deposition"]
        if fused and not dep:
            continue
        probe = LaserProbe(rec["laser_calls"])
        streams = [oa.stream_create() for _ in range(nt)]

        def body(r, w):
            st = streams[r]
            link = InProcLink(w, r, particle_capacity=256, stream=st, periodic=d["left"] == "internal") if nt > 1 else None
            g = build(oa, d, r, st)
            load(g, rec["fields"][r])
            el = make_pop(oa, oa.ELECTRON, rec["electrons_created"][r], dx, dt, stream=st)
            io = make_pop(oa, oa.ION, rec["ions_created"][r], dx, dt, cs, stream=st)
            t = fh(rec["t0"])
            for _ in range(rec["steps"]):
                if fused:
                    g.clear()
                    el.advance_deposit(g, dt, link)
                    io.advance_deposit(g, dt, link)
                else:
                    el.advance(g, dt, link)
                    io.advance(g, dt, link)
                    if dep:
                        g.clear()
                        g.deposit(el, dt)
                        g.deposit(io, dt)
                g.synchronize(probe.fn("y"), probe.fn("z"), t, dt, link)
                g.advance(dt)
                t += dt
            out = {"grid": {c: g.raw(c) for c in FIELDS}, "energy": g.em_field_energy(), "t": t,
                   "e": el.soa(prev_x=not fused), "i": io.soa(prev_x=not fused), "ke": (el.total_kinetic_energy(), io.total_kinetic_energy())}
            return out

        try:
            res = run_world(nt, body)
        finally:
            for st in streams:
                oa.stream_destroy(st)
        probe.done()
        for r, out in enumerate(res):
            what = f"{rec['name']} rank {r} fused={fused}"
            assert abs(out["t"] - fh(rec["t_end"])) <= 1e-15 * abs(fh(rec["t_end"]))
            assert [len(out["e"]["cell"]), len(out["i"]["cell"])] == list(rec["counts_after"][r]), what
            for comps in (("Ex", "Ey", "Ez"), ("Bx", "By", "Bz")):
                s = max(group_scale(g_, comps) for g_ in rec["grids"])
                for c in comps:
                    close(out["grid"][c], fh(rec["grids"][r][c]), s, TOL_RUN, f"{what}: {c}")
            w = fh(rec["em_field_energy_local"][r])
            close(out["energy"], w, abs(w), TOL_RUN, f"{what}: energy")
            for key, name in (("e", "electrons"), ("i", "ions")):
                s, want = out[key], rec[name][r]
                wc = np.array(want["cell"], dtype=np.int64)
                got = canon(s["cell"].astype(np.int64), s["x"], s["ux"], s["uy"], s["uz"])
                exp = canon(wc, *[fh(want[k]) for k in ("x", "ux", "uy", "uz")])
                assert np.array_equal(got[0], exp[0]), f"{what}: {name} cells"
                if len(wc):
                    us = max(float(np.max(np.abs(np.concatenate(exp[2:5])))), 1e-3)
                    close(got[1], exp[1], 1.0, TOL_RUN, f"{what}: {name} x")
                    for k in (2, 3, 4):
                        close(got[k], exp[k], us, TOL_RUN, f"{what}: {name} u")


def test_photon_push(oa, gold):
    """opal_pop_advance for the photon species with chi, y, z tracked (photon.rs:150-183) and its kinetic energy
    (photon.rs:219-225), three steps on random fields."""
    for rec in gold["photon_push"]:
        d = rec["design"]
        dx, dt = fh(d["dx"]), fh(rec["dt"])
        g = build(oa, d, 0)
        load(g, rec["fields"])
        ph = make_pop(oa, oa.PHOTON, rec["created"], dx, dt)
        ph.track_extras(True)
        fields = {c: fh(rec["fields"][c]) for c in FIELDS}
        e_s, b_s = max(np.abs(fields[c]).max() for c in ("Ex", "Ey", "Ez")), max(np.abs(fields[c]).max() for c in ("Bx", "By", "Bz"))
        kmax = max(np.abs(fh(rec["created"][c])).max() for c in ("ux", "uy", "uz")) * np.sqrt(3.0)
        S = (kmax * e_s + 2.997925e8 * kmax * b_s) / 1.323285e18      # the terms chi's difference of squares is made of
        for k, want in enumerate(rec["steps"]):
            ph.advance(g, dt)
            s = ph.soa()  # (prev_x is not kept for photons: they are never deposited, photon.rs:197-199)
            ex = ph.extras()
            what = f"photons, {rec['name']}, step {k}"
            wc = np.array(want["cell"], dtype=np.int64)
            got = canon(s["cell"].astype(np.int64), s["x"], s["x"], s["ux"], s["uy"], s["uz"], ex["chi"], ex["y"], ex["z"])
            exp = canon(wc, *[fh(want[key]) for key in ("x", "x", "ux", "uy", "uz", "chi", "y", "z")])
            assert np.array_equal(got[0], exp[0]), f"{what}: cells"
            close(got[1], exp[1], 1.0, TOL, f"{what}: x")
            for q in (3, 4, 5):
                assert np.array_equal(got[q], exp[q]), f"{what}: a photon's momentum never changes"
            close(got[6] ** 2, exp[6] ** 2, S * S, 8 * TOL, f"{what}: chi^2")
            ys = max(np.abs(exp[7]).max(), np.abs(exp[8]).max(), 1e-300)
            close(got[7], exp[7], ys, 4 * TOL, f"{what}: y")
            close(got[8], exp[8], ys, 4 * TOL, f"{what}: z")
            ke = fh(want["kinetic_energy_sequential_sum"])
            close(ph.total_kinetic_energy(), ke, abs(ke), 1e-11, f"{what}: KE")

