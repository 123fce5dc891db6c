"""The CPU oracle against textbook physics, where the reference holds no test of its own (gather,
deposit and the closed loop have none): the same two statements tests/test_gpu_known_answers.py makes of the HIP
path, made of oracle/opal_oracle.c.  A mis-reading of the reference that oracle and kernels share would have to
reproduce the Yee dispersion relation to 1e-8 and the plasma frequency to 0.2 % to get past both."""
import numpy as np

C = 2.997925e8            # constants.rs:4
C2_FIELD = 89875517873681764.0  # constants.rs:6: the c^2 of the field solve is the exact one
EPS0 = 8.854188e-12
E_CHARGE = 1.602177e-19
M_E = 9.109383e-31


def test_light_wave_on_the_yee_mesh(oracle):
    """yee.rs:839-867 (B half step, E step, B half step) has the dispersion relation
    sin(omega dt / 2) = (c dt / dx) sin(k dx / 2); its right-going eigenmode returns after N steps shifted by
    omega N dt, amplitude and energy intact."""
    o = oracle
    nx, dx, m = 240, 1.0e-8, 12
    dt = 0.9 * dx / C
    cf = np.sqrt(C2_FIELD)
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    k = 2.0 * np.pi * m / (nx * dx)
    w = 2.0 / dt * np.arcsin(cf * dt / dx * np.sin(0.5 * k * dx))
    j = np.arange(g.len) - L
    e0 = 3.0e10
    b0 = e0 * dx * np.sin(w * dt) / (2.0 * C2_FIELD * dt * np.sin(0.5 * k * dx))
    g.set_raw("Ey", e0 * np.sin(k * dx * j))
    g.set_raw("Bz", b0 * np.sin(k * dx * (j + 0.5)))
    g.synchronize()
    e_before = g.em_field_energy()
    steps = 800
    for _ in range(steps):
        g.advance(dt)
        g.synchronize()
    jj = np.arange(nx)
    got = g.raw("Ey")[L:L + nx]
    c_, s_ = 2.0 / nx * np.sum(got * np.cos(k * dx * jj)), 2.0 / nx * np.sum(got * np.sin(k * dx * jj))
    phase = np.arctan2(-c_, s_)
    d = (phase - (w * steps * dt) % (2.0 * np.pi) + np.pi) % (2.0 * np.pi) - np.pi
    assert abs(d) < 1e-8
    assert abs(np.hypot(c_, s_) / e0 - 1.0) < 1e-9
    assert abs((cf * k - w) * steps * dt) > 0.1  # not the continuum's c k
    assert abs(g.em_field_energy() / e_before - 1.0) < 1e-9
    want_bz = b0 * np.sin(k * dx * (jj + 0.5) - w * steps * dt)
    assert np.max(np.abs(g.raw("Bz")[L:L + nx] - want_bz)) < 1e-8 * b0
    assert np.max(np.abs(g.raw("Ez"))) == 0.0 and np.max(np.abs(g.raw("By"))) == 0.0


def test_cold_plasma_oscillates_at_the_plasma_frequency(oracle):
    """fields_at -> Electron::push -> deposit -> synchronize -> advance closed on itself: omega_p = sqrt(n e^2 / eps0 m)."""
    o = oracle
    nx, dx, npc, ne, u0 = 256, 1.0e-8, 16, 1.0e27, 1.0e-3
    dt = 0.95 * dx / C
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    cell = np.repeat(np.arange(nx), npc).astype(np.int64)
    x = np.tile((np.arange(npc) + 0.5) / npc, nx)
    X = (cell + x) * dx
    ux = u0 * np.sin(2.0 * np.pi * X / (nx * dx))
    z = np.zeros_like(x)
    wgt = np.full_like(x, ne * dx / npc)
    pe = o.Population.from_soa(o.ELECTRON, cell, x, ux, z, z, wgt, dx, dt)
    pi = o.Population.from_soa(o.ION, cell, x, z, z, z, wgt, dx, dt)
    pi.set_charge_state(1.0, 1.0, 1.0e9)  # ballast: too heavy to move
    wp = np.sqrt(ne * E_CHARGE ** 2 / (EPS0 * M_E))
    steps = int(3.6 * 2.0 * np.pi / (wp * dt))
    kx = 2.0 * np.pi * (np.arange(nx) + 0.5) / nx
    a = np.empty(steps + 1)
    t = 0.0
    for i in range(steps + 1):
        a[i] = 2.0 / nx * np.sum(g.raw("Ex")[L:L + nx] * np.sin(kx))
        if i < steps:
            assert o.world_step([g], [pe], [pi], True, t, dt) == 0
            t += dt
    amp = M_E * C * wp * u0 / E_CHARGE
    assert abs(np.max(np.abs(a)) / amp - 1.0) < 5e-3
    assert a[1] > 0.0
    s = np.sign(a[1:])
    idx = np.nonzero(s[1:] * s[:-1] < 0)[0] + 1
    tz = (idx + a[idx] / (a[idx] - a[idx + 1])) * dt
    assert len(tz) >= 6
    half_period = (tz[-1] - tz[0]) / (len(tz) - 1)
    assert abs(np.pi / half_period / wp - 1.0) < 2e-3


def test_light_in_a_cold_plasma(oracle):
    """The transverse half of the loop.  Omega^2 = c^2 K^2 + omega_p^2 Wg Wd with Omega = (2/dt) sin(omega dt/2),
    K = (2/dx) sin(k dx/2), Wg = sinc^3(k dx/2) and Wd = (1 - 1/48) sinc^3(k dx/2): the reference's transverse taps
    end in weight(2.0 + x) = 0 (yee.rs:597, 602), so 1/48 of j_y is not deposited; the amplitude of the wave,
    u0 m c omega_p^2 Wd / (e Omega cos(omega dt/2)), resolves that."""
    o = oracle
    nx, dx, npc, ne, u0, m = 256, 1.0e-8, 32, 1.0e27, 1.0e-3, 4
    dt = 0.95 * dx / C
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    cell = np.repeat(np.arange(nx), npc).astype(np.int64)
    x = np.tile((np.arange(npc) + 0.5) / npc, nx)
    uy = u0 * np.sin(2.0 * np.pi * m * (cell + x) / nx)
    z = np.zeros_like(x)
    wgt = np.full_like(x, ne * dx / npc)
    pe = o.Population.from_soa(o.ELECTRON, cell, x, z, uy, z, wgt, dx, dt)
    pi = o.Population.from_soa(o.ION, cell, x, z, z, z, wgt, dx, dt)
    pi.set_charge_state(1.0, 1.0, 1.0e9)
    wp = np.sqrt(ne * E_CHARGE ** 2 / (EPS0 * M_E))
    cf = np.sqrt(C2_FIELD)
    kdx = 2.0 * np.pi * m / nx
    sinc3 = (np.sin(0.5 * kdx) / (0.5 * kdx)) ** 3
    Wg, Wd = sinc3, (1.0 - 1.0 / 48.0) * sinc3
    w = 2.0 / dt * np.arcsin(np.sqrt((cf * dt / dx * np.sin(0.5 * kdx)) ** 2 + 0.25 * (wp * dt) ** 2 * Wg * Wd))
    w_vac = 2.0 / dt * np.arcsin(cf * dt / dx * np.sin(0.5 * kdx))
    kj = kdx * np.arange(nx)
    steps = int(4.6 * 2.0 * np.pi / (w * dt))
    a = np.empty(steps + 1)
    t = 0.0
    for i in range(steps + 1):
        a[i] = 2.0 / nx * np.sum(g.raw("Ey")[L:L + nx] * np.sin(kj))
        if i < steps:
            assert o.world_step([g], [pe], [pi], True, t, dt) == 0
            t += dt
    assert a[1] > 0.0
    s = np.sign(a[1:])
    idx = np.nonzero(s[1:] * s[:-1] < 0)[0] + 1
    tz = (idx + a[idx] / (a[idx] - a[idx + 1])) * dt
    assert len(tz) >= 8
    got = np.pi / ((tz[-1] - tz[0]) / (len(tz) - 1))
    assert abs(got / w - 1.0) < 3e-3 and abs(got / w_vac - 1.0) > 0.1
    big_omega = 2.0 / dt * np.sin(0.5 * w * dt)
    amp = u0 * M_E * C * wp * wp * Wd / (E_CHARGE * big_omega * np.cos(0.5 * w * dt))
    assert abs(np.max(np.abs(a)) / amp - 1.0) < 4e-3, np.max(np.abs(a)) / amp


def _two_cold_beams(nx, dx, m, npc, u0, eps):
    """Two cold counter-streaming electron beams, quiet start, a velocity ripple of relative size eps in mode m; the beam
    density is chosen so that mode m is the fastest-growing one: k v0 = (sqrt 3 / 2) omega_b."""
    gam = np.sqrt(1.0 + u0 * u0)
    v0 = u0 / gam * C
    k = 2.0 * np.pi * m / (nx * dx)
    wb = k * v0 / (0.5 * np.sqrt(3.0))
    nb = wb ** 2 * EPS0 * M_E * gam ** 3 / E_CHARGE ** 2   # omega_b^2 = n_b e^2 / (eps0 m gamma^3): longitudinal mass
    cell = np.repeat(np.arange(nx), npc).astype(np.int64)
    x = np.tile((np.arange(npc) + 0.5) / npc, nx)
    ripple = 1.0 + eps * np.sin(2.0 * np.pi * m * (cell + x) / nx)
    cell2, x2 = np.concatenate([cell, cell]), np.concatenate([x, x])
    ux2 = np.concatenate([u0 * ripple, -u0 * ripple])
    order = np.argsort(cell2, kind="stable")
    z = np.zeros(len(x2))
    w = np.full(len(x2), nb * dx / npc)
    return (cell2[order], x2[order], ux2[order], z, z.copy(), w), wb


def growth_rate(a, dt):
    """Slope of ln a over the stretch in which the growing mode has outrun the seed's other components (50 x the first
    steps' amplitude) and is still linear (3 % of the saturated one)."""
    lo, hi = 50.0 * np.max(a[:60]), 0.03 * np.max(a)
    i0, i1 = int(np.argmax(a > lo)), int(np.argmax(a > hi))
    assert i1 - i0 > 400, (i0, i1)
    return np.polyfit(np.arange(i0, i1) * dt, np.log(a[i0:i1]), 1)[0]


def test_two_stream_instability_grows_at_half_the_beam_plasma_frequency(oracle):
    """The workload of the headline benchmark as a known answer: for two equal cold beams
    1 = omega_b^2 [1/(omega - k v0)^2 + 1/(omega + k v0)^2] has its fastest-growing root Im omega = omega_b / 2 at
    k v0 = (sqrt 3 / 2) omega_b.  Seven e-foldings of the seeded mode's Ex amplitude, fitted: 1 %."""
    o = oracle
    nx, dx, m = 256, 4.0e-9, 3
    dt = 0.95 * dx / C
    st, wb = _two_cold_beams(nx, dx, m, 32, 0.2, 1e-7)
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    pe = o.Population.from_soa(o.ELECTRON, *st, dx, dt)
    kx = 2.0 * np.pi * m * (np.arange(nx) + 0.5) / nx
    steps = 2200
    a = np.empty(steps)
    for i in range(steps):
        pe.advance(g, dt, 4)
        g.clear()
        g.deposit(pe, dt, 4)
        g.synchronize(None, None, 0.0, dt)
        g.advance(dt)
        ex = g.raw("Ex")[L:L + nx]
        a[i] = np.hypot(np.sum(ex * np.sin(kx)), np.sum(ex * np.cos(kx))) * 2.0 / nx
    assert np.max(a) > 1e6 * a[0]
    assert abs(growth_rate(a, dt) / (0.5 * wb) - 1.0) < 1e-2


# ------------------------------------------------------------------ photon emission against synchrotron theory
ALPHA_FINE = 7.29735257e-3     # constants.rs:24
COMPTON_TIME = 1.28808867e-21  # constants.rs:26
CRITICAL_FIELD = 1.323285e18   # constants.rs:22
M_E_MEV = 0.510999             # constants.rs:20


def quantum_power_factor(chi):
    """g(chi) = radiated power / classical power of quantum synchrotron emission, by quadrature (g -> 1 for chi -> 0)."""
    from scipy import integrate, special
    f = lambda y: (2.0 * y * y * special.kv(5.0 / 3.0, y) / (2.0 + 3.0 * chi * y) ** 2
                   + 36.0 * chi * chi * y ** 3 * special.kv(2.0 / 3.0, y) / (2.0 + 3.0 * chi * y) ** 4)
    return 9.0 * np.sqrt(3.0) / (8.0 * np.pi) * integrate.quad(f, 0.0, np.inf, limit=400)[0]


import pytest  # noqa: E402


@pytest.mark.parametrize("chi0,dt,tol", [(0.005, 2.0e-16, 0.03), (0.3, 5.0e-18, 0.03)])
def test_radiated_power_is_the_synchrotron_power(oracle, chi0, dt, tol):
    """The oracle's emission chain (Electron::push's optical depth, emit_radiation: rate, spectrum, recoil) closed on
    physics: electrons of gamma = 1000 gyrating in a static B at quantum parameter chi0 radiate
    (2/3) alpha chi^2 m c^2 / tau_C times g(chi) -- g by quadrature of the quantum spectrum above chi = 0.01; below it
    the reference's own low-chi model, (1 - 8 chi / (5 sqrt 3)) (1 - (55 sqrt 3 / 48) chi) (photon_emission.rs:61-62,
    139-142) -- and the photons carry what the electrons lost.  20 000 electrons, two emissions each: 1 % statistics."""
    o = oracle
    n, gamma0, per_electron, nx = 20_000, 1000.0, 2.0, 32
    dx = 2.0 * C * dt
    g = o.Grid.single(nx, 0.0, dx)
    g.set_raw("Bz", np.full(g.len, chi0 * CRITICAL_FIELD / (gamma0 * C)))
    rng = np.random.default_rng(2)
    cell = np.sort(rng.integers(0, nx, n)).astype(np.int64)
    z = np.zeros(n)
    p = o.Population.from_soa(o.ELECTRON, cell, rng.random(n), np.full(n, -np.sqrt(gamma0 ** 2 - 1.0)), z, z, np.ones(n), dx, dt)
    p.set_electron_state(tau=rng.exponential(size=n), ids=np.arange(n, dtype=np.uint64))
    ph = o.Population.empty(o.PHOTON)
    ke0 = p.total_kinetic_energy()
    if chi0 < 0.01:
        factor = lambda chi: (1.0 - 8.0 * chi / (5.0 * np.sqrt(3.0))) * (1.0 - 55.0 * np.sqrt(3.0) / 48.0 * chi)
    else:
        lc = np.linspace(np.log(chi0 / 100.0), np.log(chi0 * 1.01), 32)
        gg = np.array([quantum_power_factor(np.exp(v)) for v in lc])
        factor = lambda chi: np.interp(np.log(np.maximum(chi, 1e-300)), lc, gg)
    mc2 = M_E_MEV * 1.0e6 * E_CHARGE
    w0 = o.rate(chi0, gamma0)
    assert w0 * dt < 0.02
    steps = int(per_electron / (w0 * dt))
    energy = 0.0
    for s in range(steps):
        p.advance(g, dt, 4)
        chi = p.electron_extras()["chi"]
        if s == 0:
            assert abs(chi.mean() / chi0 - 1.0) < 2e-5
        energy += np.sum((2.0 / 3.0) * ALPHA_FINE * chi ** 2 * mc2 / COMPTON_TIME * factor(chi)) * dt
        o.emit_radiation(p, ph, seed=5, step=s, t=s * dt)
    lost, radiated = ke0 - p.total_kinetic_energy(), ph.total_kinetic_energy()
    assert abs(radiated / lost - 1.0) < 1e-5
    assert len(ph) > 1.6 * n
    assert abs(radiated / energy - 1.0) < tol, radiated / energy


@pytest.mark.parametrize("umax", [0.3, 20.0])
def test_the_oracles_deposit_satisfies_the_continuity_equation(oracle, umax):
    """Cell::flux / deposit (yee.rs:185-204, 551-641) have no test in the reference, so the oracle's reading of them is
    checked against something that does not depend on any reading: with the charge of a macro-particle spread as a
    triangle of half-width one cell, (rho_i(X_new) - rho_i(X_old)) / dt + (jx[i] - jx[i-1]) / dx = 0 cell by cell, rho
    computed in numpy straight from the cumulative triangle (the statement tests/test_gpu_fullsize.py makes of the HIP
    deposit).  Relativistic electrons and C6+ in random fields, periodic, every tap of the stencil live."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_fullsize import QE, continuity_residual, node_charge
    o = oracle
    nx, npc, dx = 96, 100, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(5)
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    for comp, amp in (("Ex", 2e12), ("Ey", 2e12), ("Ez", 2e12), ("Bx", 5e3), ("By", 5e3), ("Bz", 5e3)):
        g.set_raw(comp, amp * rng.standard_normal(g.len))
    g.synchronize(None, None, 0.0, 0.0)
    pops = []
    for species, zstar, amass, seed in ((o.ELECTRON, 1.0, 1.0, 1), (o.ION, 6.0, 12.0, 2)):
        r = np.random.default_rng(seed)
        n = nx * npc
        cell = np.repeat(np.arange(nx, dtype=np.int64), npc)
        u = umax * r.standard_normal((3, n))
        p = o.Population.from_soa(species, cell, r.random(n), u[0], u[1], u[2], 1.0e3 * (0.5 + r.random(n)), dx, dt)
        if species == o.ION:
            p.set_charge_state(zstar, zstar, amass)
        pops.append((p, QE if species == o.ELECTRON else -QE * zstar))
    for step in range(3):
        before = [(p.soa(), q) for p, q in pops]
        for p, _ in pops:
            p.advance(g, dt, 2)
        g.clear()
        for p, _ in pops:
            g.deposit(p, dt, 2)
        g.synchronize(None, None, 0.0, dt)
        jx = g.raw("jx")[L:L + nx]
        rho_old = sum(node_charge(s["cell"], s["x"], q * s["w"], nx) for s, q in before)
        rho_new = sum(node_charge(s["cell"], s["x"], q * s["w"], nx) for s, q in [(p.soa(), q) for p, q in pops])
        res = continuity_residual(rho_old, rho_new, jx, dt, dx)
        scale = abs(QE) * 1.0e3 * npc * 7 / (dt * dx)
        assert np.max(np.abs(res)) < 1e-12 * scale, (step, np.max(np.abs(res)) / scale)
        assert np.max(np.abs(jx)) > 1e-3 * scale * dx
        g.advance(dt)


# ------------------------------------------------------------------ the operators the reference holds no test for
M_P = 1.672622e-27        # constants.rs:18


def test_the_oracles_gather_is_a_quadratic_spline_at_the_documented_stagger(oracle):
    """fields_at (yee.rs:499-529): a linear field comes back exactly and x^2 as x_p^2 + 1/4 when the values are placed
    where yee.rs:70-82 says each component lives (Ex, By, Bz at cell centres; Ey, Ez at left-hand edges; Bx un-interpolated)."""
    o = oracle
    nx, dx = 64, 1.0
    g = o.Grid.single(nx, 0.0, dx)
    L = g.left_bdy_size
    j = np.arange(g.len, dtype=np.float64)
    rng = np.random.default_rng(11)
    coef = {k: (rng.normal(), rng.normal()) for k in ("Ex", "Ey", "Ez", "By", "Bz")}
    centred = {"Ex": True, "By": True, "Bz": True, "Ey": False, "Ez": False}
    for k, (a, b) in coef.items():
        g.set_raw(k, a + b * (j + (0.5 if centred[k] else 0.0)))
    bx = rng.normal(size=g.len)
    g.set_raw("Bx", bx)
    for c, xi in zip(rng.integers(0, nx, 300), np.concatenate([[0.0, 0.5, 0.499999999, 0.999999999], rng.random(296)])):
        E, B = g.fields_at(int(c), float(xi))
        pos = c + L + xi
        got = {"Ex": E[0], "Ey": E[1], "Ez": E[2], "By": B[1], "Bz": B[2]}
        for k, (a, b) in coef.items():
            assert abs(got[k] - (a + b * pos)) <= 2e-13 * max(abs(a + b * pos), 1.0), (k, c, xi)
        assert B[0] == bx[c + L]
    g.set_raw("Ex", (j + 0.5) ** 2)
    g.set_raw("Ey", j ** 2)
    for c, xi in zip(rng.integers(0, nx, 100), rng.random(100)):
        E, _ = g.fields_at(int(c), float(xi))
        pos = c + L + xi
        assert abs(E[0] - (pos ** 2 + 0.25)) <= 1e-12 * pos ** 2 and abs(E[1] - (pos ** 2 + 0.25)) <= 1e-12 * pos ** 2


@pytest.mark.parametrize("axis,z_star,a_mass", [("z", 1.0, 1.0), ("z", 6.0, 12.0), ("x", 1.0, 1.0)])
def test_the_oracles_ion_push_is_the_boris_rotation(oracle, axis, z_star, a_mass):
    """Ion::push (ion.rs:168-214): |u| conserved, u turned by 2 atan(alpha c |B| / gamma) per step about B in the sense
    of a positive charge -- the analogue, for ions, of the reference's electron.rs:401-421."""
    o = oracle
    nx, steps = 64, 100
    q, M = z_star * E_CHARGE, a_mass * M_P
    u0 = np.array([0.8, 0.3, 0.0]) if axis == "z" else np.array([0.1, 0.7, -0.4])
    gamma = np.sqrt(1.0 + u0 @ u0)
    b0 = 2.0
    dt = (2.0 * np.pi / 60.0) * gamma * M / (q * b0)
    dx = 3.0 * C * dt
    g = o.Grid.single(nx, 0.0, dx)
    g.set_raw("Bz" if axis == "z" else "Bx", np.full(g.len, b0))
    n = 64
    rng = np.random.default_rng(13)
    cell = np.sort(rng.integers(0, nx, n)).astype(np.int64)
    p = o.Population.from_soa(o.ION, cell, rng.random(n), np.full(n, u0[0]), np.full(n, u0[1]), np.full(n, u0[2]), np.ones(n), dx, dt)
    p.set_charge_state(z_star, z_star, a_mass)
    for _ in range(steps):
        p.advance(g, dt)
    s = p.soa()
    alpha = q * dt / (2.0 * M * C)
    theta = 2.0 * np.arctan(alpha * C * b0 / gamma)
    u = np.stack([s["ux"], s["uy"], s["uz"]], axis=1)
    assert np.max(np.abs(np.sqrt(np.sum(u * u, axis=1)) - np.sqrt(u0 @ u0))) < 1e-12
    if axis == "z":
        phi = np.arctan2(u0[1], u0[0]) - steps * theta
        want = np.array([np.hypot(u0[0], u0[1]) * np.cos(phi), np.hypot(u0[0], u0[1]) * np.sin(phi), 0.0])
    else:
        phi = np.arctan2(u0[2], u0[1]) - steps * theta
        want = np.array([u0[0], np.hypot(u0[1], u0[2]) * np.cos(phi), np.hypot(u0[1], u0[2]) * np.sin(phi)])
    assert np.max(np.abs(u - want)) < 1e-10
