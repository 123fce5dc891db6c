"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, plus size-independent properties at the BASELINE sizes.

Tolerances (all f64): single-operator parity 1e-12 relative to the scale of the quantity (the only
differences from the oracle are FMA contraction, reciprocal-multiply for the loop-invariant
divisions by dt, dx and gamma, and summation order); N-step field parity 1e-9 over tens of steps
(north_star asks 1e-6).
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C = 2.997925e8
EPS0 = 8.854188e-12
QE = -1.602177e-19


@pytest.fixture(scope="module")
def oa():
    import opal_amd
    assert opal_amd.device_count() >= 1
    return opal_amd


def _laser_setup():
    xmin = -10.0e-6
    dx = 1.0e-6 / 100.0
    dt = 0.95 * dx / C
    omega = 2.0 * math.pi * C / 1.0e-6
    return xmin, dx, dt, omega


def _cmp_grids(g, og, comps, tol):
    for comp in comps:
        a, b = g.raw(comp), og.raw(comp)
        scale = max(np.max(np.abs(b)), 1e-300)
        err = np.max(np.abs(a - b)) / scale
        assert err < tol, (comp, err)


FIELDS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz")


def test_laser_bc(oa, oracle):
    """yee.rs:903-953 on the GPU, cell by cell against the oracle and against the analytic energy."""
    o = oracle
    xmin, dx, dt, omega = _laser_setup()
    t = -15.0e-6 / C
    nsteps = int(abs(t) / dt)
    emax = 1.0e10

    def laser_y(t, z):
        phi = omega * (t - z / C)
        return emax * math.sin(phi) * math.cos(phi / 16.0) ** 2 if abs(phi) < 8.0 * math.pi else 0.0

    g = oa.Grid.single(2000, xmin, dx, oa.BDY_LASER, oa.BDY_ABSORBING)
    og = o.Grid.single(2000, xmin, dx, o.BDY_LASER, o.BDY_ABSORBING)
    assert (g.len, g.left_bdy_size, g.right_bdy_size) == (og.len, og.left_bdy_size, og.right_bdy_size)
    for _ in range(nsteps + 1):
        g.synchronize(laser_y, lambda t, x: 0.0, t, dt)
        g.advance(dt)
        og.synchronize(laser_y, None, t, dt)
        og.advance(dt)
        t += dt
    _cmp_grids(g, og, ("Ey", "Bz"), 1e-10)
    target = EPS0 * emax ** 2 * (3.0 * math.pi * C / omega)
    assert abs(g.em_field_energy() - target) / target < 1.0e-3
    assert abs(g.em_field_energy() - og.em_field_energy()) / target < 1e-11
    e, _ = g.fields_at(1000 + 25, 0.0)
    eo, _ = og.fields_at(1000 + 25, 0.0)
    assert abs(e[0, 1] - eo[1]) / emax < 1e-10


def test_conducting_bc(oa, oracle):
    """yee.rs:956-1011"""
    o = oracle
    xmin, dx, dt, omega = _laser_setup()
    t = -15.0e-6 / C
    nsteps = int(35.0e-6 / (C * dt))
    emax = 1.0e10

    def laser_y(t, z):
        phi = omega * (t - z / C)
        if abs(phi) < 8.0 * math.pi:
            return -emax * (math.sin(phi) + (2.0 / 16.0) * math.cos(phi) * math.tan(phi / 16.0)) * math.cos(phi / 16.0) ** 2
        return 0.0

    def laser_z(t, z):
        phi = omega * (t - z / C)
        if abs(phi) < 8.0 * math.pi:
            return emax * (math.cos(phi) + (2.0 / 16.0) * math.sin(phi) * math.tan(phi / 16.0)) * math.cos(phi / 16.0) ** 2
        return 0.0

    g = oa.Grid.single(2000, xmin, dx, oa.BDY_LASER, oa.BDY_CONDUCTING)
    og = o.Grid.single(2000, xmin, dx, o.BDY_LASER, o.BDY_CONDUCTING)
    for _ in range(nsteps + 1):
        g.synchronize(laser_y, laser_z, t, dt)
        g.advance(dt)
        og.synchronize(laser_y, laser_z, t, dt)
        og.advance(dt)
        t += dt
    _cmp_grids(g, og, ("Ey", "Ez", "By", "Bz"), 1e-10)
    target = EPS0 * emax ** 2 * (193.0 / 32.0) * (math.pi * C / omega)
    assert abs(g.em_field_energy() - target) / target < 1.0e-3


def test_periodic_bc(oa, oracle):
    """yee.rs:1014-1079"""
    o = oracle
    xmin, dx, dt, omega = _laser_setup()
    t = -5.0e-6 / C
    nsteps = int(10.0e-6 / (C * dt))
    jmax = omega * EPS0 * 1.0e5
    g = oa.Grid.single(2000, xmin, dx)
    og = o.Grid.single(2000, xmin, dx)
    assert g.to_left() == 0 and g.to_right() == 0

    def run(t):
        for _ in range(nsteps):
            phi = omega * t
            v = 0.0
            if abs(phi) < 4.0 * math.pi:
                v = jmax * (math.sin(phi) + (2.0 / 8.0) * math.cos(phi) * math.tan(phi / 8.0)) * math.cos(phi / 8.0) ** 2
            for gg in (g, og):
                jz = gg.raw("jz")
                jz[1000] = v
                gg.set_raw("jz", jz)
                gg.synchronize(None, None, t, dt)
                gg.advance(dt)
            t += dt
        return t

    t = run(t)
    e0 = g.em_field_energy()
    t = run(t)
    e1 = g.em_field_energy()
    assert e0 > 0 and abs(e0 - e1) / e0 < 1.0e-6
    _cmp_grids(g, og, ("Ez", "By", "jz"), 1e-10)


def _random_grid_pair(oa, o, nx, dx, seed, left=0, right=0, amp_e=1e3, amp_b=1e-5):
    rng = np.random.default_rng(seed)
    g = oa.Grid.single(nx, 0.0, dx, left, right)
    og = o.Grid.single(nx, 0.0, dx, left, right)
    for comp in FIELDS:
        amp = amp_e if comp[0] == "E" else amp_b
        a = amp * rng.standard_normal(g.len)
        g.set_raw(comp, a)
        og.set_raw(comp, a)
    if left == 0 and right == 0:
        # ghost cells hold the periodic images of the interior, as they do in any run after the first
        # synchronize (yee.rs:424-452); the HIP gather may read either image of a cell
        g.synchronize(None, None, 0.0, 0.0)
        og.synchronize(None, None, 0.0, 0.0)
    return g, og


def test_fields_at(oa, oracle):
    """yee.rs:499-529 with random fields and offsets, incl. the ends of the domain and xi = 0."""
    g, og = _random_grid_pair(oa, oracle, 64, 500.0, 3)
    rng = np.random.default_rng(4)
    c = np.concatenate([rng.integers(0, 64, 500), [0, 0, 63, 63]]).astype(np.int32)
    xi = np.concatenate([rng.random(500), [0.0, 0.999999, 0.0, 0.5]])
    E, B = g.fields_at(c, xi)
    for k in range(len(c)):
        eo, bo = og.fields_at(int(c[k]), float(xi[k]))
        assert np.max(np.abs(E[k] - eo)) <= 1e-13 * 1e3 * 4
        assert np.max(np.abs(B[k] - bo)) <= 1e-13 * 1e-5 * 4
    with pytest.raises(oa.OpalError):
        g.fields_at([64], [0.1])  # yee.rs:502 assert


def _random_particles(nx, npc, dx, seed, umax):
    rng = np.random.default_rng(seed)
    n = nx * npc
    cell = np.repeat(np.arange(nx, dtype=np.int32), npc)
    x = rng.random(n)
    u = umax * rng.standard_normal((3, n))
    # unique weights double as particle identities
    w = 1.0e3 * (1.0 + np.arange(n) * 1e-9)
    return cell, x, u[0], u[1], u[2], w


def _by_id(d):
    order = np.argsort(d["w"], kind="stable")
    return {k: v[order] for k, v in d.items()}


@pytest.mark.parametrize("species,umax,zstar,amass", [(0, 0.5, 1.0, 1.0), (0, 30.0, 1.0, 1.0),
                                                      (1, 0.05, 6.0, 12.0), (1, 0.5, 1.0, 9.109383e-31 / 1.672622e-27)])
def test_single_step_push(oa, oracle, species, umax, zstar, amass):
    """Population::advance, one step from identical state (Vay: electron.rs:268-330; Boris: ion.rs:168-214)."""
    o = oracle
    nx, npc, dx = 48, 37, 1.0e-8
    dt = 0.95 * dx / C
    g, og = _random_grid_pair(oa, o, nx, dx, 11, amp_e=3e12, amp_b=1e4)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 12, umax)
    p = oa.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(0)
    op = o.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    if species == 1:
        p.set_charge_state(zstar, zstar, amass)
        op.set_charge_state(zstar, zstar, amass)
    p.advance(g, dt)
    assert op.advance(og, dt) == 0
    a, b = _by_id(p.soa(prev_x=True)), _by_id(op.soa())
    assert len(a["w"]) == len(b["w"]) == nx * npc  # periodic: nobody is lost
    assert np.array_equal(a["cell"], b["cell"].astype(np.int32))
    for k in ("x", "prev_x"):
        assert np.max(np.abs(a[k] - b[k])) < 1e-12, k
    for k in ("ux", "uy", "uz"):
        scale = np.max(np.abs(b[k]))
        assert np.max(np.abs(a[k] - b[k])) / scale < 1e-12, k
    ke, oke = p.total_kinetic_energy(), op.total_kinetic_energy()
    assert abs(ke - oke) / oke < 1e-12


def _deposit_scale(w, charge, dx, dt, nx):
    # largest current a cell can see: every particle of a cell crossing a whole cell in dt
    return np.sum(np.abs(w)) * abs(charge) / dt / nx


@pytest.mark.parametrize("species", [0, 1])
def test_deposit_after_advance(oa, oracle, species):
    """YeeGrid::deposit (yee.rs:551-641) on the state left by advance: jx, jy, jz and rho incl. quirks."""
    o = oracle
    nx, npc, dx = 40, 53, 1.0e-8
    dt = 0.95 * dx / C
    g, og = _random_grid_pair(oa, o, nx, dx, 21, amp_e=1e12, amp_b=3e3)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 22, 1.5)
    p = oa.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(0)
    op = o.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    p.advance(g, dt)
    op.advance(og, dt)
    g.clear()
    og.clear()
    g.deposit(p, dt)
    assert og.deposit(op, dt, nthreads=3) == 0
    # the HIP deposit may use either periodic image of a cell (ghost or far end); they agree after the fold
    g.synchronize(None, None, 0.0, dt)
    og.synchronize(None, None, 0.0, dt)
    q = abs(QE)
    sj = _deposit_scale(w, q, dx, dt, nx)
    for comp in ("jx", "jy", "jz"):
        assert np.max(np.abs(g.raw(comp) - og.raw(comp))) < 1e-12 * sj, comp
    srho = np.sum(w) * q / dx / nx
    assert np.max(np.abs(g.raw("rho") - og.raw("rho"))) < 1e-12 * srho
    # size-independent properties: total charge and total current (sum of flux = displacement)
    assert abs(np.sum(g.raw("rho")) - np.sum(og.raw("rho"))) < 1e-12 * srho * nx


@pytest.mark.parametrize("species", [0, 1])
def test_fused_equals_operator_sequence(oa, oracle, species):
    """The fused pass must equal advance -> clear -> deposit of the reference's loop (main.rs:239-257)."""
    o = oracle
    nx, npc, dx = 64, 41, 1.0e-8
    dt = 0.95 * dx / C
    g, og = _random_grid_pair(oa, o, nx, dx, 31, amp_e=1e12, amp_b=3e3)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 32, 2.0)
    p = oa.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(0)
    op = o.Population.from_soa(species, cell, x, ux, uy, uz, w, dx, dt)
    g.clear()
    p.advance_deposit(g, dt)
    op.advance(og, dt)
    og.clear()
    og.deposit(op, dt)
    # a particle that wrapped around deposits into the ghost cells on the fused path and into the
    # far end of the domain in the reference's order; the two agree once synchronize has folded the ghosts
    g.synchronize(None, None, 0.0, dt)
    og.synchronize(None, None, 0.0, dt)
    sj = _deposit_scale(w, abs(QE), dx, dt, nx)
    for comp in ("jx", "jy", "jz"):
        assert np.max(np.abs(g.raw(comp) - og.raw(comp))) < 1e-12 * sj, comp
    srho = np.sum(w) * abs(QE) / dx / nx
    assert np.max(np.abs(g.raw("rho") - og.raw("rho"))) < 1e-12 * srho
    a, b = _by_id(p.soa()), _by_id(op.soa())
    assert np.array_equal(a["cell"], b["cell"].astype(np.int32))
    assert np.max(np.abs(a["x"] - b["x"])) < 1e-12


def _two_stream_state(nx, npc, dx, seed):
    rng = np.random.default_rng(seed)
    n = nx * npc
    ud = 2.5e-24 / (9.109383e-31 * C)
    cell = np.repeat(np.arange(nx, dtype=np.int32), npc)
    x = rng.random(n)
    ux = ud * (1.0 + 0.001 * rng.standard_normal(n)) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    uy = np.zeros(n)
    uz = np.zeros(n)
    w = np.full(n, 20.0 * dx / npc)
    return cell, x, ux, uy, uz, w


def test_wrapped_particles_stay_in_their_tile(oa, oracle):
    """One periodic subdomain: particles that wrap around sit in slots sorted for the other end of the array.  The end
    chunks' tiles are modular (tile cell t = interior cell (base + t) mod size), so 30 steps after a re-binning fast
    counter-streaming beams still have no tile miss, and fields and energies follow the oracle (main.rs:238-267)."""
    o = oracle
    nx, npc, dx = 128, 1024, 500.0  # two cells per 2048-slot chunk, as dense runs have
    dt = 0.95 * dx / C
    rng = np.random.default_rng(77)
    n = nx * npc
    cell = np.repeat(np.arange(nx, dtype=np.int32), npc)
    x = rng.random(n)
    ux = 0.4 * (1.0 + 0.01 * rng.standard_normal(n)) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    z = np.zeros(n)
    w = np.full(n, 20.0 * dx / npc)
    g = oa.Grid.single(nx, 0.0, dx)
    og = o.Grid.single(nx, 0.0, dx)
    pe = oa.Population.from_soa(oa.ELECTRON, cell, x, ux, z, z, w, dx, dt)
    pi = oa.Population.from_soa(oa.ION, cell, x, -ux, z, z, w, dx, dt)
    a_pos = 9.109383e-31 / 1.672622e-27
    pi.set_charge_state(1.0, 1.0, a_pos)
    ope = o.Population.from_soa(o.ELECTRON, cell, x, ux, z, z, w, dx, dt)
    opi = o.Population.from_soa(o.ION, cell, x, -ux, z, z, w, dx, dt)
    opi.set_charge_state(1.0, 1.0, a_pos)
    for p in (pe, pi):
        p.sort(g)
        p.set_sort_interval(0)
    t = 0.0
    for step in range(30):
        g.clear()
        pe.advance_deposit(g, dt)
        pi.advance_deposit(g, dt)
        g.synchronize(None, None, t, dt)
        g.advance(dt)
        assert o.world_step([og], [ope], [opi], True, t, dt, nthreads=2) == 0
        t += dt
    moved = pe.soa()
    assert np.count_nonzero(moved["cell"][: 2 * npc] > nx // 2) > npc // 4  # the first slots did wrap around
    for p in (pe, pi):
        assert p.counters()["tile_misses"] == 0
    _cmp_grids(g, og, ("Ex",), 1e-9)
    for p, op in ((pe, ope), (pi, opi)):
        ke, oke = p.total_kinetic_energy(), op.total_kinetic_energy()
        assert abs(ke - oke) / oke < 1e-12
        assert len(p) == len(op) == n


@pytest.mark.parametrize("fused", [True, False])
def test_two_stream_steps(oa, oracle, fused):
    """40 steps of main.rs:238-267 (electrons + positrons-as-ions, periodic) against the oracle."""
    o = oracle
    nx, npc, dx = 128, 100, 500.0
    dt = 0.95 * dx / C
    g = oa.Grid.single(nx, 0.0, dx)
    og = o.Grid.single(nx, 0.0, dx)
    se = _two_stream_state(nx, npc, dx, 41)
    si = _two_stream_state(nx, npc, dx, 42)
    pe = oa.Population.from_soa(oa.ELECTRON, *se, dx, dt)
    pi = oa.Population.from_soa(oa.ION, *si, dx, dt)
    a_pos = 9.109383e-31 / 1.672622e-27
    pi.set_charge_state(1.0, 1.0, a_pos)
    ope = o.Population.from_soa(o.ELECTRON, *se, dx, dt)
    opi = o.Population.from_soa(o.ION, *si, dx, dt)
    opi.set_charge_state(1.0, 1.0, a_pos)
    t = 0.0
    for step in range(40):
        if fused:
            g.clear()
            pe.advance_deposit(g, dt)
            pi.advance_deposit(g, dt)
        else:
            pe.advance(g, dt)
            pi.advance(g, dt)
            g.clear()
            g.deposit(pe, dt)
            g.deposit(pi, dt)
        g.synchronize(None, None, t, dt)
        g.advance(dt)
        assert o.world_step([og], [ope], [opi], True, t, dt, nthreads=2) == 0
        t += dt
    _cmp_grids(g, og, ("Ex",), 1e-9)
    fe, ofe = g.em_field_energy(), og.em_field_energy()
    assert abs(fe - ofe) / ofe < 1e-9
    for p, op in ((pe, ope), (pi, opi)):
        ke, oke = p.total_kinetic_energy(), op.total_kinetic_energy()
        assert abs(ke - oke) / oke < 1e-12
        assert len(p) == len(op) == nx * npc


def test_open_boundaries_delete_leavers(oa, oracle):
    """Leavers at Laser/Absorbing edges are deleted and deposit nothing (yee.rs:329-343, main.rs:239-257)."""
    o = oracle
    nx, npc, dx = 32, 20, 1.0e-8
    dt = 0.95 * dx / C
    g = oa.Grid.single(nx, 0.0, dx, oa.BDY_LASER, oa.BDY_ABSORBING)
    og = o.Grid.single(nx, 0.0, dx, o.BDY_LASER, o.BDY_ABSORBING)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 52, 3.0)
    p = oa.Population.from_soa(oa.ELECTRON, cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(0)
    op = o.Population.from_soa(o.ELECTRON, cell, x, ux, uy, uz, w, dx, dt)
    for _ in range(6):
        g.clear()
        p.advance_deposit(g, dt)
        op.advance(og, dt)
        og.clear()
        og.deposit(op, dt)
        assert len(p) == len(op)
        sj = _deposit_scale(w, abs(QE), dx, dt, nx)
        assert np.max(np.abs(g.raw("jx") - og.raw("jx"))) < 1e-12 * sj
    assert len(p) < nx * npc
    a, b = _by_id(p.soa()), _by_id(op.soa())
    assert np.array_equal(a["w"], b["w"])
    assert np.array_equal(a["cell"], b["cell"].astype(np.int32))


def test_sort_rebins_without_changing_the_population(oa):
    nx, npc, dx = 256, 30, 500.0
    dt = 0.95 * dx / C
    g = oa.Grid.single(nx, 0.0, dx)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 61, 0.3)
    perm = np.random.default_rng(0).permutation(len(cell))
    p = oa.Population.from_soa(oa.ELECTRON, cell[perm], x[perm], ux[perm], uy[perm], uz[perm], w[perm], dx, dt)
    before = _by_id(p.soa())
    p.sort(g)
    after = p.soa()
    assert np.all(np.diff(after["cell"]) >= 0)  # sortedness
    after = _by_id(after)
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    p.sort(g)  # idempotent on the cell sequence
    assert np.array_equal(p.soa()["cell"], np.sort(cell))


def test_invariant_violation_is_reported(oa):
    """A particle moving >= 1 cell trips the reference's assert (electron.rs:313) -> OPAL_ERR_ASSERT."""
    nx, dx = 16, 1.0e-8
    dt = 1.2 * dx / C  # deliberately beyond the CFL-limited dt
    g = oa.Grid.single(nx, 0.0, dx)
    n = 64
    p = oa.Population.from_soa(oa.ELECTRON, np.full(n, 5, np.int32), np.full(n, 0.5), np.full(n, 100.0),
                               np.zeros(n), np.zeros(n), np.ones(n), dx, dt)
    p.advance(g, dt)
    with pytest.raises(oa.OpalError) as e:
        p.total_kinetic_energy()
    assert e.value.status == 3


# ----------------------------------------------------------- BASELINE-size property tests
@pytest.mark.parametrize("npc", [2441])
def test_full_size_properties(oa, npc):
    """configs[1] shape (4096 cells, ~1e7 particles per species): size-independent properties of the
    fused pass -- particle count conserved under periodic wrap, sum(rho) dx = total charge,
    sum(jx) dx = sum Q * displacement / dt (charge-conserving flux), fused == unfused."""
    nx, dx = 4096, 500.0
    dt = 0.95 * dx / C
    cell, x, ux, uy, uz, w = _two_stream_state(nx, npc, dx, 71)
    n = len(cell)
    g = oa.Grid.single(nx, 0.0, dx)
    p = oa.Population.from_soa(oa.ELECTRON, cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(0)
    g.clear()
    p.advance_deposit(g, dt)
    jx_raw, rho_raw = g.raw("jx"), g.raw("rho")
    g.synchronize(None, None, 0.0, dt)
    jx_f, rho_f = g.raw("jx"), g.raw("rho")
    after = p.soa()
    assert len(after["cell"]) == n
    assert after["cell"].min() >= 0 and after["cell"].max() < nx
    assert np.all((after["x"] >= 0.0) & (after["x"] < 1.0))
    q = QE * w[0]
    assert abs(np.sum(rho_raw) * dx - q * n) < 1e-9 * abs(q * n)
    # displacement in cells (periodic unwrap)
    d = (after["cell"].astype(np.int64) - cell) + (after["x"] - x)
    d = np.where(d > nx / 2, d - nx, np.where(d < -nx / 2, d + nx, d))
    total = q * np.sum(d) / dt
    scale = abs(q) * np.sum(np.abs(d)) / dt
    assert abs(np.sum(jx_raw) - total) < 1e-9 * scale
    # unfused path from the same initial state
    g2 = oa.Grid.single(nx, 0.0, dx)
    p2 = oa.Population.from_soa(oa.ELECTRON, cell, x, ux, uy, uz, w, dx, dt)
    p2.set_sort_interval(0)
    p2.advance(g2, dt)
    g2.clear()
    g2.deposit(p2, dt)
    g2.synchronize(None, None, 0.0, dt)
    sj = abs(q) * npc / dt
    assert np.max(np.abs(g2.raw("jx") - jx_f)) < 1e-11 * sj
    assert np.max(np.abs(g2.raw("rho") - rho_f)) < 1e-11 * abs(q) * npc / dx
    ke, ke2 = p.total_kinetic_energy(), p2.total_kinetic_energy()
    assert abs(ke - ke2) / ke < 1e-13


def test_extras_survive_rebinning(oa, oracle):
    """chi, work, y, z (electron.rs:18-25) follow their particle through the counting sort."""
    o = oracle
    nx, npc, dx = 48, 30, 1.0e-8
    dt = 0.95 * dx / C
    g, og = _random_grid_pair(oa, o, nx, dx, 81, amp_e=1e12, amp_b=3e3)
    cell, x, ux, uy, uz, w = _random_particles(nx, npc, dx, 82, 1.0)
    p = oa.Population(oa.ELECTRON, 4 * nx * npc)
    p.track_extras(True)
    p.upload(cell, x, ux, uy, uz, w, dx, dt)
    p.set_sort_interval(2)
    op = o.Population.from_soa(o.ELECTRON, cell, x, ux, uy, uz, w, dx, dt)
    for _ in range(5):
        p.advance(g, dt)
        op.advance(og, dt)
    a = p.soa()
    ex = p.extras()
    order = np.argsort(a["w"], kind="stable")
    ref = op.soa()
    rex = op.electron_extras()
    rorder = np.argsort(ref["w"], kind="stable")
    assert p.counters()["steps_since_sort"] < 2  # it did re-bin
    for k in ("chi", "work", "y", "z"):
        scale = np.max(np.abs(rex[k]))
        assert np.max(np.abs(ex[k][order] - rex[k][rorder])) <= 1e-10 * scale, k


def test_rebinning_of_a_nearly_ordered_population_keeps_every_particle_whole(oa):
    """The re-binning as a run meets it (particles.hip: k_sort_hist, k_sort_scan, k_sort_scatter_staged): slots sorted for a
    cell their particle has since drifted up to 13 cells from (a wave holds ~20 distinct cells: counts and ranks in the LDS
    window, the scatter staged through LDS), cold cells that fill whole chunks (every lane of a wave in ONE cell: the wave's
    shared increment), cells thousands of particles deep, a few hundred far-away cells (wrapped particles: outside the window,
    straight to their slot), prev_x and the emission state (tau, id) carried.  Every particle must come out with all of its
    attributes, in cell order."""
    nx, dx = 4096, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(123)
    g = oa.Grid.single(nx, 0.0, dx)
    hot_home = np.repeat(np.arange(1000, 2000, dtype=np.int64), 60)                  # 60 000 hot particles
    hot = hot_home + rng.integers(-13, 14, len(hot_home))
    cold = np.repeat(np.arange(2500, 2600, dtype=np.int64), 300)                     # 30 000 cold ones: 300 per cell
    deep = np.repeat(np.array([3000, 3001], dtype=np.int64), 5000)                   # two cells 5000 deep (several chunks each)
    cell = np.concatenate([hot, cold, deep])
    far = rng.choice(len(cell), 300, replace=False)
    cell[far] = rng.integers(0, nx, 300)                                             # strays anywhere
    n = len(cell)
    x = rng.random(n)
    u = rng.standard_normal((3, n))
    w = 1.0e3 * (1.0 + np.arange(n) * 1e-9)
    p = oa.Population(oa.ELECTRON, n + 4096)
    p.track_extras(2)
    p.upload(cell.astype(np.int32), x, u[0], u[1], u[2], w, dx, dt)
    tau = rng.random(n)
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919)) ^ np.uint64(0xABCDEF)
    p.set_emission_state(tau, ids)
    before = p.soa(prev_x=True)
    p.sort(g)
    after = p.soa(prev_x=True)
    t_after, i_after = p.emission_state()
    assert len(after["w"]) == n and np.all(np.diff(after["cell"]) >= 0)
    assert np.array_equal(np.bincount(after["cell"], minlength=nx), np.bincount(cell, minlength=nx))
    order = np.argsort(after["w"], kind="stable")
    for k in ("cell", "x", "ux", "uy", "uz", "w", "prev_x"):
        assert np.array_equal(after[k][order], before[k]), k   # (uploaded in w order: `before` is already by identity)
    assert np.array_equal(t_after[order], tau) and np.array_equal(i_after[order], ids)


def test_rebinning_with_the_scatter_straight_to_the_output(oa):
    """OPAL_SORT_STAGED=0 keeps round 3's scatter (every lane stores to the run of its own cell) as the other leg of the A/B
    in profiles/r04_ab/ab_sort_staged.txt; the switch is read once per process, so the same test runs in a child."""
    import subprocess
    import sys
    env = dict(os.environ, OPAL_SORT_STAGED="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "nearly_ordered or without_changing_the_population or extras_survive"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
