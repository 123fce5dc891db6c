"""Known answers for the parts of the hot path the reference holds no test for (`fields_at`'s
values, `Ion::push`), checked on the HIP path WITHOUT the oracle: every expected value below is an analytic
property of the algorithm the reference's source states, so a mis-reading shared by oracle/opal_oracle.c and
pic_device.hpp cannot pass here.

* gather (yee.rs:499-529, stagger documented at yee.rs:70-82): the quadratic B-spline reproduces a linear field
  exactly and a quadratic one up to its constant second moment 1/4 -- evaluated at the positions the reference
  documents (Ex, By, Bz at cell centres, Ey, Ez at left-hand edges, Bx taken from the particle's cell as is);
* ion push (ion.rs:168-214): the Boris rotation conserves |u| exactly and turns u by 2 atan(alpha c |B| / gamma)
  per step about B, in the sense of a positive charge; in a static E the momentum grows by 2 alpha E per step --
  the analogues of electron.rs:401-436 for `OPAL_ION`, through `opal_pop_advance`.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C = 2.997925e8            # constants.rs:4
E_CHARGE = 1.602177e-19   # constants.rs:12-14
M_P = 1.672622e-27        # constants.rs:18
M_E = 9.109383e-31        # constants.rs:16
M_E_MEV = 0.510999        # constants.rs:20


@pytest.fixture(scope="module")
def oa():
    import opal_amd
    return opal_amd


def test_gather_reproduces_linear_fields_at_the_documented_stagger(oa):
    nx, dx = 96, 1.0e-7
    g = oa.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    j = np.arange(g.len, dtype=np.float64)
    rng = np.random.default_rng(11)
    coef = {k: (rng.normal(), rng.normal()) for k in ("Ex", "Ey", "Ez", "By", "Bz")}
    centred = {"Ex": True, "By": True, "Bz": True, "Ey": False, "Ez": False}
    for k, (a, b) in coef.items():
        g.set_raw(k, a + b * (j + (0.5 if centred[k] else 0.0)))  # the value AT the point the component lives at
    bx = rng.normal(size=g.len)
    g.set_raw("Bx", bx)
    n = 4000
    c = rng.integers(0, nx, n).astype(np.int32)
    xi = rng.random(n)
    xi[:4] = [0.0, 0.5, 0.499999999, 0.999999999]  # the spline's knots
    E, B = g.fields_at(c, xi)
    pos = c + L + xi  # the particle's position in units of dx, from the left end of the array
    got = {"Ex": E[:, 0], "Ey": E[:, 1], "Ez": E[:, 2], "By": B[:, 1], "Bz": B[:, 2]}
    for k, (a, b) in coef.items():
        want = a + b * pos
        assert np.max(np.abs(got[k] - want)) <= 2e-13 * np.max(np.abs(want)), k
    assert np.array_equal(B[:, 0], bx[c + L])  # yee.rs:519: Bx is not interpolated


def test_gather_is_a_quadratic_spline(oa):
    """f(x) = x^2 comes back as x_p^2 + 1/4: the second moment of the order-2 B-spline, independent of where in the
    cell the particle sits (a linear/CIC weighting would give x_p^2 + xi (1 - xi))."""
    nx, dx = 64, 1.0
    g = oa.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    j = np.arange(g.len, dtype=np.float64)
    g.set_raw("Ex", (j + 0.5) ** 2)
    g.set_raw("Ey", j ** 2)
    g.set_raw("Bz", (j + 0.5) ** 2)
    rng = np.random.default_rng(12)
    c = rng.integers(0, nx, 1000).astype(np.int32)
    xi = rng.random(1000)
    E, B = g.fields_at(c, xi)
    pos = c + L + xi
    for got in (E[:, 0], E[:, 1], B[:, 2]):
        assert np.max(np.abs(got - (pos ** 2 + 0.25))) <= 1e-12 * np.max(pos ** 2)


@pytest.mark.parametrize("axis,z_star,a_mass", [("z", 1.0, 1.0), ("z", 6.0, 12.0), ("x", 1.0, 1.0), ("z", 1.0, M_E / M_P)])
def test_ion_boris_rotation_in_a_static_magnetic_field(oa, axis, z_star, a_mass):
    nx, steps = 64, 100
    q, M = z_star * E_CHARGE, a_mass * M_P
    u0 = np.array([0.8, 0.3, 0.0]) if axis == "z" else np.array([0.1, 0.7, -0.4])
    gamma = np.sqrt(1.0 + u0 @ u0)
    # choose dt for ~1/60 of a gyration per step in B0, and dx so that the ion crosses a third of a cell per step
    b0 = 2.0
    dt = (2.0 * np.pi / 60.0) * gamma * M / (q * b0)
    dx = 3.0 * C * dt
    g = oa.Grid.single(nx, 0.0, dx)
    g.set_raw("Bz" if axis == "z" else "Bx", np.full(g.len, b0))
    n = 512
    rng = np.random.default_rng(13)
    cell = np.sort(rng.integers(0, nx, n)).astype(np.int32)
    x = rng.random(n)
    p = oa.Population.from_soa(oa.ION, cell, x, np.full(n, u0[0]), np.full(n, u0[1]), np.full(n, u0[2]), np.ones(n), dx, dt)
    p.set_charge_state(z_star, z_star, a_mass)
    for _ in range(steps):
        p.advance(g, dt)
    s = p.soa()
    alpha = q * dt / (2.0 * M * C)                       # ion.rs:173-175
    theta = 2.0 * np.arctan(alpha * C * b0 / gamma)      # Boris: tan(theta / 2) = t |cB|, t = alpha / gamma
    u = np.stack([s["ux"], s["uy"], s["uz"]], axis=1)
    assert np.max(np.abs(np.sqrt(np.sum(u * u, axis=1)) - np.sqrt(u0 @ u0))) < 1e-12  # gamma conserved
    if axis == "z":
        # a positive charge turns clockwise about +z: phi decreases by theta per step
        phi = np.arctan2(u0[1], u0[0]) - steps * theta
        up = np.hypot(u0[0], u0[1])
        want = np.array([up * np.cos(phi), up * np.sin(phi), 0.0])
    else:
        # ... and clockwise about +x in the (y, z) plane; ux untouched
        phi = np.arctan2(u0[2], u0[1]) - steps * theta
        up = np.hypot(u0[1], u0[2])
        want = np.array([u0[0], up * np.cos(phi), up * np.sin(phi)])
    assert np.max(np.abs(u - want)) < 1e-10
    if axis == "z":
        # positions: x advances with the NEW velocity each step (ion.rs:197-199)
        k = np.arange(1, steps + 1)
        moved = np.sum(C * up * np.cos(np.arctan2(u0[1], u0[0]) - k * theta) * dt / (gamma * dx))
        order = np.argsort(s["w"], kind="stable")  # (all weights equal: slot order is kept by an un-rebinned advance)
        have = (s["cell"] + s["x"]) % nx
        want_pos = (cell + x + moved) % nx
        d = np.abs(np.sort(have) - np.sort(want_pos))
        assert np.max(np.minimum(d, nx - d)) < 1e-9, order[:1]


def test_ion_in_a_static_electric_field(oa):
    """ion.rs:176-196 with B = 0: u_x grows by 2 alpha E per step exactly (the two half kicks; the rotation is the
    identity); the kinetic energy follows ion.rs:128-134."""
    nx, steps = 32, 100
    z_star, a_mass = 6.0, 12.0
    q, M = z_star * E_CHARGE, a_mass * M_P
    e0 = 1.0e12
    dt = 1.0e-17
    dx = 1.0e-6
    g = oa.Grid.single(nx, 0.0, dx)
    g.set_raw("Ex", np.full(g.len, e0))
    n = 256
    cell = np.repeat(np.arange(nx), n // nx).astype(np.int32)
    x = np.linspace(0.05, 0.95, n)
    w = np.full(n, 3.0)
    p = oa.Population.from_soa(oa.ION, cell, x, np.zeros(n), np.zeros(n), np.zeros(n), w, dx, dt)
    p.set_charge_state(z_star, z_star, a_mass)
    for _ in range(steps):
        p.advance(g, dt)
    s = p.soa()
    alpha = q * dt / (2.0 * M * C)
    want = 2.0 * alpha * e0 * steps
    assert want > 1e-4
    assert np.max(np.abs(s["ux"] - want)) <= 1e-12 * want
    assert np.all(s["uy"] == 0.0) and np.all(s["uz"] == 0.0)
    gamma_m1 = want * want / (1.0 + np.sqrt(1.0 + want * want))
    ke = np.sum(w) * gamma_m1 * (a_mass * M_P / M_E) * M_E_MEV * 1.0e6 * E_CHARGE
    assert abs(p.total_kinetic_energy() - ke) <= 1e-11 * ke
    # gamma(t) = sqrt(1 + (q E t / M c)^2), the form of electron.rs:423-436
    assert abs(np.sqrt(1.0 + s["ux"][0] ** 2) - np.sqrt(1.0 + (q * e0 * steps * dt / (M * C)) ** 2)) < 1e-12


# ------------------------------------------------------------------ the whole loop against textbook physics
EPS0 = 8.854188e-12       # constants.rs:8

LANGMUIR = """---
control:
 dx: 1.0e-8
 nx: {nx}
 xmin: 0.0
 start: 0.0
 end: 1.0e-12
 current_deposition: true
 n_outputs: 2
 balance: false
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: 64
 ne: {ne}
 ux: {ux}
 uy: {uy}
 uz: 0.0
 output: x:px
ions:
 name: ballast
 npc: 4
 Z: 1.0
 A: 1.0e9
 ni: {ne}
 ux: 0.0
 uy: 0.0
 uz: 0.0
 output: x:px
"""


@pytest.mark.parametrize("fused", [True, False])
def test_cold_plasma_oscillates_at_the_plasma_frequency(oa, fused):
    """gather -> push -> deposit -> synchronize -> advance, closed on itself: a cold electron fluid given the velocity
    u0 sin(kx) over ions too heavy to move (A = 1e9; Ex = 0 at t = 0, and Ampere's law alone evolves it -- which is what
    a charge-conserving deposit is for) rings at omega_p = sqrt(n e^2 / eps0 m), whatever k.  The leapfrog correction
    sin(omega dt / 2) = omega_p dt / 2 is 1e-4 here and the spline's form factor (k dx)^2 ~ 6e-4: the measured frequency
    must agree with the textbook one to 0.2 %.  Nothing of the oracle enters."""
    nx, ne, u0 = 256, 1.0e27, 1.0e-3
    sim = oa.Simulation(yaml_text=LANGMUIR.format(nx=nx, ne=ne, ux=f"{u0} * sin(2.0 * pi * x / ({nx} * 1.0e-8))", uy="0.0"))
    sim.set_fused(fused)
    g = sim.grid
    L = g.left_bdy_size
    dt = sim.dt
    wp = np.sqrt(ne * E_CHARGE ** 2 / (EPS0 * M_E))
    assert 0.04 < wp * dt < 0.07
    kx = 2.0 * np.pi * (np.arange(nx) + 0.5) / nx  # Ex lives at cell centres (yee.rs:70-82)
    steps = int(3.6 * 2.0 * np.pi / (wp * dt))
    a = np.empty(steps + 1)
    b = np.empty(steps + 1)
    for i in range(steps + 1):
        ex = g.raw("Ex")[L:L + nx]
        a[i] = 2.0 / nx * np.sum(ex * np.sin(kx))
        b[i] = 2.0 / nx * np.sum(ex * np.cos(kx))
        if i < steps:
            sim.step(1)
    sim.sync()
    # E(x, t) = +(m c omega_p u0 / e) sin(kx) sin(omega_p t): amplitude, phase and shape
    amp = M_E * C * wp * u0 / E_CHARGE
    assert abs(np.max(np.abs(a)) / amp - 1.0) < 5e-3
    assert np.max(np.abs(b)) < 2e-3 * amp           # the mode stays a sine: nothing drifts
    assert a[1] > 0.0                                # electrons moving to +x carry j < 0: dEx/dt = -j / eps0 > 0 there
    # frequency from the zero crossings of the mode amplitude (linear interpolation between the bracketing steps)
    s = np.sign(a[1:])
    idx = np.nonzero(s[1:] * s[:-1] < 0)[0] + 1
    tz = (idx + a[idx] / (a[idx] - a[idx + 1])) * dt
    assert len(tz) >= 6
    half_period = (tz[-1] - tz[0]) / (len(tz) - 1)
    assert abs(np.pi / half_period / wp - 1.0) < 2e-3
    # energy sloshes between the field and the electrons and the sum stays put
    fe, ke = sim.energies()[0], sim.energies()[1]
    n_macro = nx * 64
    ke0 = 0.25 * u0 ** 2 * M_E * C ** 2 * ne * nx * 1.0e-8  # <sin^2> = 1/2, per unit area
    # (momenta and fields are not known at the same instant, so the sum read at one instant wobbles by
    # O(omega_p dt) = 5 %; the amplitude check above is the sharp statement that all of ke0 reaches the field)
    assert abs((fe + ke) / ke0 - 1.0) < 0.04, (fe, ke, ke0, n_macro)


def test_light_in_a_cold_plasma(oa):
    """The transverse half of the loop (gather of Ey, the u_y kick, the deposit of j_y, Ampere's law with both the curl
    and the current): electrons given u_y = u0 sin(kx) and no field radiate a standing wave whose Ey rings at
    omega^2 = omega_p^2 + c^2 k^2 -- on the mesh, with Omega = (2/dt) sin(omega dt/2) and K = (2/dx) sin(k dx/2),
        Omega^2 = c^2 K^2 + omega_p^2 Wg Wd,     Ey = u0 m c omega_p^2 Wd / (e Omega cos(omega dt/2)) sin(kx) sin(omega t),
    Wg = sinc^3(k dx/2) the quadratic spline of the gather and Wd = (1 - 1/48) sinc^3(k dx/2) that of the deposit --
    the reference's j_y, j_z taps (yee.rs:594-597, 599-602) end in weight(2.0 + x), identically zero, where the spline's
    fourth tap would be weight(2.0 - x): a particle in the right half of its cell deposits 1 - (x - 1/2)^2 / 2 of its
    transverse current, 1 - 1/48 on average.  Frequency to 0.3 % (the plasma term is a quarter of omega^2: without it
    13 % lower) and amplitude to 0.4 % -- which resolves the 2 % of that tap, so a `corrected' deposit fails here."""
    nx, ne, u0, m = 256, 1.0e27, 1.0e-3, 4
    dx = 1.0e-8
    sim = oa.Simulation(yaml_text=LANGMUIR.format(nx=nx, ne=ne, ux="0.0",
                                                  uy=f"{u0} * sin(2.0 * pi * {m} * x / ({nx} * 1.0e-8))"))
    g = sim.grid
    L = g.left_bdy_size
    dt = sim.dt
    wp = np.sqrt(ne * E_CHARGE ** 2 / (EPS0 * M_E))
    cf = np.sqrt(89875517873681764.0)
    kdx = 2.0 * np.pi * m / nx
    sinc3 = (np.sin(0.5 * kdx) / (0.5 * kdx)) ** 3
    Wg, Wd = sinc3, (1.0 - 1.0 / 48.0) * sinc3
    w_vac = 2.0 / dt * np.arcsin(cf * dt / dx * np.sin(0.5 * kdx))
    w = 2.0 / dt * np.arcsin(np.sqrt((cf * dt / dx * np.sin(0.5 * kdx)) ** 2 + 0.25 * (wp * dt) ** 2 * Wg * Wd))
    assert 0.2 < (w * w - w_vac * w_vac) / (w * w) < 0.3
    kj = kdx * np.arange(nx)  # Ey lives at the left-hand edges (yee.rs:70-82)
    steps = int(4.6 * 2.0 * np.pi / (w * dt))
    a = np.empty(steps + 1)
    for i in range(steps + 1):
        a[i] = 2.0 / nx * np.sum(g.raw("Ey")[L:L + nx] * np.sin(kj))
        if i < steps:
            sim.step(1)
    sim.sync()
    assert a[1] > 0.0  # j_y < 0 where the electrons move to +y
    s = np.sign(a[1:])
    idx = np.nonzero(s[1:] * s[:-1] < 0)[0] + 1
    tz = (idx + a[idx] / (a[idx] - a[idx + 1])) * dt
    assert len(tz) >= 8
    half_period = (tz[-1] - tz[0]) / (len(tz) - 1)
    got = np.pi / half_period
    assert abs(got / w - 1.0) < 3e-3, (got, w, w_vac)
    assert abs(got / w_vac - 1.0) > 0.1
    # (the zero-frequency part of the start state -- a static Bz held by the residual current -- carries no E)
    big_omega = 2.0 / dt * np.sin(0.5 * w * dt)
    amp = u0 * M_E * C * wp * wp * Wd / (E_CHARGE * big_omega * np.cos(0.5 * w * dt))
    assert abs(np.max(np.abs(a)) / amp - 1.0) < 4e-3, np.max(np.abs(a)) / amp
    # (Ex is second order in u0: the v_y x B_z force)
    assert np.max(np.abs(g.raw("Ex"))) < 2e-3 * amp and np.max(np.abs(g.raw("Ez"))) == 0.0


def test_light_wave_on_the_yee_mesh(oa):
    """synchronize (periodic wrap) + advance alone, against the dispersion relation of the scheme the reference states
    (yee.rs:839-867: B half step, E step, B half step -- Yee 1966 with B also known at the integer times):
    sin(omega dt / 2) = (c dt / dx) sin(k dx / 2).  The right-going eigenmode of that scheme,
        Ey_j^n = E0 sin(k x_j - omega t_n),   Bz_{j+1/2}^n = E0 dx sin(omega dt) / (2 c^2 dt sin(k dx / 2)) sin(k x_{j+1/2} - omega t_n),
    must come back after N steps with its amplitude, shifted by omega N dt -- measurably less than c k N dt -- and with
    nothing in Ez / By.  No oracle.  (The c of the field solve is sqrt(SPEED_OF_LIGHT_SQD) = 299 792 458 m/s,
    constants.rs:6, not the rounded SPEED_OF_LIGHT of constants.rs:4 the pushes use: 800 steps tell the two apart.)"""
    nx, dx, m = 240, 1.0e-8, 12
    dt = 0.9 * dx / C
    c2 = 89875517873681764.0
    cf = np.sqrt(c2)
    g = oa.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    k = 2.0 * np.pi * m / (nx * dx)
    w = 2.0 / dt * np.arcsin(cf * dt / dx * np.sin(0.5 * k * dx))
    j = np.arange(g.len) - L
    e0 = 3.0e10
    b0 = e0 * dx * np.sin(w * dt) / (2.0 * c2 * dt * np.sin(0.5 * k * dx))
    assert 0.99 < b0 * C / e0 < 1.0
    g.set_raw("Ey", e0 * np.sin(k * dx * j))            # Ey at the left-hand edge of cell j (yee.rs:70-82)
    g.set_raw("Bz", b0 * np.sin(k * dx * (j + 0.5)))    # Bz at its centre
    g.synchronize()
    e_before = g.em_field_energy()
    steps = 800
    for _ in range(steps):
        g.advance(dt)
        g.synchronize()
    got = g.raw("Ey")[L:L + nx]
    jj = np.arange(nx)
    c_, s_ = 2.0 / nx * np.sum(got * np.cos(k * dx * jj)), 2.0 / nx * np.sum(got * np.sin(k * dx * jj))
    # Ey = e0 sin(k x - w t) = e0 [sin kx cos wt - cos kx sin wt]
    phase = np.arctan2(-c_, s_)
    want = (w * steps * dt) % (2.0 * np.pi)
    d = (phase - want + np.pi) % (2.0 * np.pi) - np.pi
    assert abs(d) < 1e-8, (phase, want)
    w_rounded_c = 2.0 / dt * np.arcsin(C * dt / dx * np.sin(0.5 * k * dx))
    assert abs((w_rounded_c - w) * steps * dt) > 1e-5
    assert abs(np.hypot(c_, s_) / e0 - 1.0) < 1e-9
    assert abs((cf * k - w) * steps * dt) > 0.1          # a wave at c would sit 0.17 rad further on
    assert abs(g.em_field_energy() / e_before - 1.0) < 1e-9
    assert np.max(np.abs(g.raw("Ez"))) == 0.0 and np.max(np.abs(g.raw("By"))) == 0.0
    resid = got - (s_ * np.sin(k * dx * jj) + c_ * np.cos(k * dx * jj))
    assert np.max(np.abs(resid)) < 1e-9 * e0            # still the one harmonic
    bz = g.raw("Bz")[L:L + nx]
    want_bz = b0 * np.sin(k * dx * (jj + 0.5) - w * steps * dt)
    assert np.max(np.abs(bz - want_bz)) < 1e-8 * b0


@pytest.mark.parametrize("fused", [True, False])
def test_two_stream_instability_grows_at_half_the_beam_plasma_frequency(oa, fused):
    """The headline workload as a known answer (no oracle): two equal cold beams,
    1 = omega_b^2 [1/(omega - k v0)^2 + 1/(omega + k v0)^2], fastest-growing root Im omega = omega_b / 2 at
    k v0 = (sqrt 3 / 2) omega_b with omega_b^2 = n_b e^2 / (eps0 m gamma^3).  The seeded mode's Ex amplitude is followed
    through seven e-foldings and the fitted rate must be the textbook one to 1 % (the CPU oracle measures 0.9995 of it:
    tests/test_oracle_physics.py)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_physics import _two_cold_beams, growth_rate
    nx, dx, m = 256, 4.0e-9, 3
    dt = 0.95 * dx / C
    st, wb = _two_cold_beams(nx, dx, m, 32, 0.2, 1e-7)
    g = oa.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    pe = oa.Population.from_soa(oa.ELECTRON, st[0].astype(np.int32), *st[1:], dx, dt)
    kx = 2.0 * np.pi * m * (np.arange(nx) + 0.5) / nx
    steps = 2200
    a = np.empty(steps)
    for i in range(steps):
        if fused:
            g.clear()
            pe.advance_deposit(g, dt)
        else:
            pe.advance(g, dt)
            g.clear()
            g.deposit(pe, dt)
        g.synchronize(None, None, 0.0, dt)
        g.advance(dt)
        ex = g.raw("Ex")[L:L + nx]
        a[i] = np.hypot(np.sum(ex * np.sin(kx)), np.sum(ex * np.cos(kx))) * 2.0 / nx
    assert np.max(a) > 1e6 * a[0]
    assert abs(growth_rate(a, dt) / (0.5 * wb) - 1.0) < 1e-2
    assert pe.counters()["tile_misses"] >= 0 and len(pe) == 2 * 32 * nx


# ------------------------------------------------------------------ photon emission against synchrotron theory
ALPHA_FINE = 7.29735257e-3     # constants.rs:24
COMPTON_TIME = 1.28808867e-21  # constants.rs:26
CRITICAL_FIELD = 1.323285e18   # constants.rs:22


def _quantum_power_factor(chi):
    """g(chi) = radiated power / classical power of quantum synchrotron emission (Baier & Katkov; e.g. Ridgers et al.,
    J. Comput. Phys. 260 (2014) eq. 6), by quadrature: g -> 1 for chi -> 0."""
    from scipy import integrate, special
    f = lambda y: (2.0 * y * y * special.kv(5.0 / 3.0, y) / (2.0 + 3.0 * chi * y) ** 2
                   + 36.0 * chi * chi * y ** 3 * special.kv(2.0 / 3.0, y) / (2.0 + 3.0 * chi * y) ** 4)
    return 9.0 * np.sqrt(3.0) / (8.0 * np.pi) * integrate.quad(f, 0.0, np.inf, limit=400)[0]


@pytest.mark.parametrize("chi0,dt,tol", [(0.005, 1.0e-16, 0.01), (0.1, 1.0e-17, 0.015), (1.0, 2.5e-18, 0.015)])
def test_radiated_power_is_the_synchrotron_power(oa, chi0, dt, tol):
    """The emission chain closed on physics (no oracle): 1e5 electrons of gamma = 1000 gyrating in a static B with quantum
    parameter chi0 -- push (chi from the fields, electron.rs:282-288; optical depth run down), emit_radiation (rate,
    spectrum, recoil, new optical depth), three emissions each.  What must come out:
    * chi on the first step is gamma |v x B| / E_crit;
    * the photons carry what the electrons lost (B does no work), to 1 / gamma^2;
    * the number of photons is the time integral of the rate over the electrons' actual chi and gamma (Poisson: 0.2 %),
      less the second emissions a step cannot hold (W dt / 2 <= 0.8 %);
    * the radiated energy is the time integral of the synchrotron power (2/3) alpha chi^2 m c^2 / tau_C times g(chi):
      g by quadrature of the quantum spectrum for chi >= 0.01; below that the reference pairs the first-order rate
      1 - 8 chi / (5 sqrt 3) (photon_emission.rs:61-62) with the classical spectrum mapped u -> u / (1 + u)
      (photon_emission.rs:139-142, 264-292), which takes <u^2> / <u> = (55 sqrt 3 / 48) chi off the mean photon energy
      (4 sqrt 3 / 15) chi gamma m c^2: its g is (1 - 0.924 chi)(1 - 1.985 chi), not the exact 1 - 5.95 chi, and that is what
      is asked for (measured: 0.9864 of the classical power at chi = 0.005, model 0.9855)."""
    n, gamma0, per_electron = 100_000, 1000.0, 3.0
    nx = 64
    dx = 2.0 * C * dt
    g = oa.Grid.single(nx, 0.0, dx)
    g.set_raw("Bz", np.full(g.len, chi0 * CRITICAL_FIELD / (gamma0 * C)))
    rng = np.random.default_rng(1)
    cell = np.sort(rng.integers(0, nx, n)).astype(np.int32)
    z = np.zeros(n)
    p = oa.Population(oa.ELECTRON, n)
    p.track_extras(2)
    p.upload(cell, rng.random(n), np.full(n, -np.sqrt(gamma0 ** 2 - 1.0)), z, z, np.ones(n), dx, dt)
    p.set_emission_state(rng.exponential(size=n), np.arange(n, dtype=np.uint64))
    ph = oa.Population(oa.PHOTON, 1024)
    ph.track_extras(1)
    ke0 = p.total_kinetic_energy()
    if chi0 < 0.01:
        factor = lambda chi: (1.0 - 8.0 * chi / (5.0 * np.sqrt(3.0))) * (1.0 - 55.0 * np.sqrt(3.0) / 48.0 * chi)
    else:
        lc = np.linspace(np.log(chi0 / 100.0), np.log(chi0 * 1.01), 48)
        gg = np.array([_quantum_power_factor(np.exp(v)) for v in lc])
        factor = lambda chi: np.interp(np.log(np.maximum(chi, 1e-300)), lc, gg)
    mc2 = M_E_MEV * 1.0e6 * E_CHARGE
    w0 = float(oa.qed_rate(np.array([chi0]), np.array([gamma0]))[0])
    assert w0 * dt < 0.016
    steps = int(per_electron / (w0 * dt))
    energy = count = 0.0
    for s in range(steps):
        p.advance(g, dt)
        chi = p.extras()["chi"]
        if s == 0:
            assert abs(chi.mean() / chi0 - 1.0) < 2e-5 and chi.std() < 1e-6 * chi0
        u = p.soa()
        gam = np.sqrt(1.0 + u["ux"] ** 2 + u["uy"] ** 2 + u["uz"] ** 2)
        energy += np.sum((2.0 / 3.0) * ALPHA_FINE * chi ** 2 * mc2 / COMPTON_TIME * factor(chi)) * dt
        count += np.sum(oa.qed_rate(chi, gam)) * dt
        oa.emit_radiation(p, ph, seed=5, step=s, t=s * dt, count=False)
    lost, radiated = ke0 - p.total_kinetic_energy(), ph.total_kinetic_energy()
    assert abs(radiated / lost - 1.0) < 1e-5
    assert len(ph) > 2.5 * n
    assert -0.012 < len(ph) / count - 1.0 < 0.004, len(ph) / count
    assert abs(radiated / energy - 1.0) < tol, radiated / energy
