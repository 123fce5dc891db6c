"""Pins the CPU oracle against the reference's own known-answer tests.

Each test restates one `#[test]` of the reference (cited), with the same
parameters and the same tolerance.  These are the only reference-held answers
for the hot path.
"""
import math

import numpy as np
import pytest

C = 2.997925e8
EPS0 = 8.854188e-12
QE = -1.602177e-19
ME = 9.109383e-31


def _laser_setup():
    xmin = -10.0e-6
    dx = 1.0e-6 / 100.0
    dt = 0.95 * dx / C
    omega = 2.0 * math.pi * C / 1.0e-6
    return xmin, dx, dt, omega


def test_laser_bc(oracle):
    """yee.rs:903-953"""
    o = oracle
    xmin, dx, dt, omega = _laser_setup()
    t = -15.0e-6 / C
    nsteps = int(abs(t) / dt)
    emax = 1.0e10

    def laser_y(t, z):
        phi = omega * (t - z / C)
        return emax * math.sin(phi) * math.cos(phi / 16.0) ** 2 if abs(phi) < 8.0 * math.pi else 0.0

    g = o.Grid.single(2000, xmin, dx, o.BDY_LASER, o.BDY_ABSORBING)
    assert g.left_bdy_size == 4 and g.right_bdy_size == 200 and g.len == 2204
    for _ in range(nsteps + 1):
        g.synchronize(laser_y, None, t, dt)
        g.advance(dt)
        t += dt
    e, _ = g.fields_at(1000 + 25, 0.0)
    assert abs(e[1] / emax - (-math.cos(math.pi / 32.0) ** 2)) < 2e-2
    e, _ = g.fields_at(1000 + 125, 0.0)
    assert abs(e[1] / emax - (-math.cos(5.0 * math.pi / 32.0) ** 2)) < 2e-2
    target = EPS0 * emax ** 2 * (3.0 * math.pi * C / omega)
    assert abs(g.em_field_energy() - target) / target < 1.0e-3


def test_conducting_bc(oracle):
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

    g = o.Grid.single(2000, xmin, dx, o.BDY_LASER, o.BDY_CONDUCTING)
    for _ in range(nsteps + 1):
        g.synchronize(laser_y, laser_z, t, dt)
        g.advance(dt)
        t += dt
    target = EPS0 * emax ** 2 * (193.0 / 32.0) * (math.pi * C / omega)
    assert abs(g.em_field_energy() - target) / target < 1.0e-3


def test_periodic_bc(oracle):
    """yee.rs:1014-1079"""
    o = oracle
    xmin, dx, dt, omega = _laser_setup()
    t = -5.0e-6 / C
    nsteps = int(10.0e-6 / (C * dt))
    jmax = omega * EPS0 * 1.0e5
    g = o.Grid.single(2000, xmin, dx)
    assert g.to_left() == 0 and g.to_right() == 0

    def run(t):
        for _ in range(nsteps):
            phi = omega * t
            jz = g.raw("jz")
            if abs(phi) < 4.0 * math.pi:
                jz[1000] = jmax * (math.sin(phi) + (2.0 / 8.0) * math.cos(phi) * math.tan(phi / 8.0)) * math.cos(phi / 8.0) ** 2
            else:
                jz[1000] = 0.0
            g.set_raw("jz", jz)
            g.synchronize(None, None, t, dt)
            g.advance(dt)
            t += dt
        return t

    t = run(t)
    e0 = g.em_field_energy()
    t = run(t)
    e1 = g.em_field_energy()
    assert e0 > 0.0
    assert abs(e0 - e1) / e0 < 1.0e-6


def test_static_magnetic_field(oracle):
    """electron.rs:401-418"""
    o = oracle
    b0, u0 = 1.0, 1.0
    r_c = ME * C * u0 / (abs(QE) * b0)
    omega_c = C * u0 / (math.sqrt(1.0 + u0 * u0) * r_c)
    dt = 0.01 * 2.0 * math.pi / omega_c
    e = o.electron_create(0, 0.0, [u0, 0.0, 0.0], 1.0, 1.0, dt)
    for _ in range(100):
        o.electron_push(e, [0.0, 0.0, 0.0], [0.0, 0.0, b0], 1.0, dt)
    target = o.electron_create(0, 0.0, [1.0, 0.0, 0.0], 1.0, 1.0, dt)
    assert abs(e.gamma - target.gamma) / target.gamma < 1.0e-6
    # a full gyro-period returns the momentum (second-order accurate)
    assert abs(e.u[0] - 1.0) < 1e-2 and abs(e.u[1]) < 5e-2


def test_static_electric_field(oracle):
    """electron.rs:424-436"""
    o = oracle
    e0, dt = 1.0e3, 1.0e-9
    e = o.electron_create(0, 0.0, [0.0, 0.0, 0.0], 1.0, 1.0, dt)
    for _ in range(100):
        o.electron_push(e, [e0, 0.0, 0.0], [0.0, 0.0, 0.0], 1.0, dt)
    target = math.sqrt(1.0 + (QE * e0 * 1.0e-7 / (ME * C)) ** 2)
    assert abs(e.gamma - target) / target < 1.0e-6


@pytest.mark.parametrize("chi,target", [(0.026, 2.07935e14), (3.5, 1.58485e16), (9.98, 3.45844e16),
                                        (12.4, 4.04647e16), (403.0, 4.46834e17)])
def test_rate(oracle, chi, target):
    """photon_emission.rs:299-336"""
    value = oracle.rate(chi, 1000.0)
    assert abs((value - target) / target) < 1.0e-3


@pytest.mark.parametrize("threads", [1, 4])
def test_comparison_sort_mode_is_a_sort_and_changes_no_physics(oracle, threads):
    """Sort mode 1 (the restatement of rayon's par_sort_unstable_by_key used by bench.py's CPU baseline: pattern-defeating
    quicksort, OpenMP tasks for the joins) against the counting sort of mode 0: after a push the particles are ordered by
    cell, nobody is lost or duplicated, and a world run gives the same fields (summation order aside)."""
    import numpy as np
    o = oracle
    C0 = 2.997925e8
    nx, npc, dx = 256, 200, 1.0e-8
    dt = 0.95 * dx / C0
    rng = np.random.default_rng(21)
    n = nx * npc
    cell = np.repeat(np.arange(nx, dtype=np.int64), npc)
    state = (cell, rng.random(n), 0.6 * rng.standard_normal(n), 0.05 * rng.standard_normal(n), 0.05 * rng.standard_normal(n),
             1.0e3 * (1.0 + 1e-9 * np.arange(n)))
    res = {}
    for mode in (0, 1):
        o.set_sort_mode(mode)
        try:
            g = o.Grid.single(nx, 0.0, dx)
            p = o.Population.from_soa(o.ELECTRON, *state, dx, dt)
            t = 0.0
            for _ in range(6):
                assert o.world_step([g], [p], None, True, t, dt, nthreads=threads) == 0
                s = p.soa()
                assert np.all(np.diff(s["cell"]) >= 0)          # sorted by cell after every step (mod.rs:288)
                t += dt
            res[mode] = (p.soa(), g.raw("Ex"), g.raw("Ey"))
        finally:
            o.set_sort_mode(0)
    a, b = res[0][0], res[1][0]
    ia, ib = np.argsort(a["w"], kind="stable"), np.argsort(b["w"], kind="stable")
    assert np.array_equal(a["w"][ia], b["w"][ib]) and len(a["w"]) == n      # the same particles
    assert np.array_equal(a["cell"][ia], b["cell"][ib])
    assert np.max(np.abs(a["x"][ia] - b["x"][ib])) < 1e-12
    for k in (1, 2):
        assert np.max(np.abs(res[0][k] - res[1][k])) <= 1e-11 * np.max(np.abs(res[0][k]))
