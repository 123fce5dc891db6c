"""The degenerate inputs of the hot path, HIP against the oracle through the C ABI: empty and one-particle
populations, ragged sizes (not a multiple of the wavefront, the workgroup or the chunk), the smallest subdomain
GridDesign admits (8 cells, grid/mod.rs:95), every particle in ONE cell, slots uploaded in random order, particles
sitting exactly on cell boundaries, a population that grows past its capacity.  Loops follow main.rs:238-267.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C = 2.997925e8
A_POS = 9.109383e-31 / 1.672622e-27
FIELDS = ("Ex", "Ey", "Ez", "By", "Bz")


@pytest.fixture(scope="module")
def oa():
    import opal_amd
    assert opal_amd.device_count() >= 1
    return opal_amd


def _pair(oa, o, species, state, dx, dt, charge=None, sort_interval=None):
    p = oa.Population.from_soa(species, *state, dx, dt)
    op = o.Population.from_soa(species, *state, dx, dt)
    if charge:
        p.set_charge_state(*charge)
        op.set_charge_state(*charge)
    if sort_interval is not None:
        p.set_sort_interval(sort_interval)
    return p, op


def _steps(oa, o, g, og, pops, opops, dt, n, fused=True):
    t = 0.0
    for _ in range(n):
        if fused:
            g.clear()
            for p in pops:
                p.advance_deposit(g, dt)
        else:
            for p in pops:
                p.advance(g, dt)
            g.clear()
            for p in pops:
                g.deposit(p, dt)
        g.synchronize(None, None, t, dt)
        g.advance(dt)
        for op in opops:
            op.advance(og, dt)
        og.clear()
        for op in opops:
            og.deposit(op, dt)
        og.synchronize(None, None, t, dt)
        og.advance(dt)
        t += dt


def _cmp_fields(g, og, tol=1e-9):
    for k in FIELDS:
        a, b = g.raw(k), og.raw(k)
        scale = max(np.max(np.abs(og.raw(c))) for c in FIELDS)
        assert np.max(np.abs(a - b)) <= tol * max(scale, 1e-300), k


def _cmp_pops(p, op, tol=1e-9):
    a, b = p.soa(), op.soa()
    assert len(a["x"]) == len(b["x"])
    ia, ib = np.argsort(a["w"], kind="stable"), np.argsort(b["w"], kind="stable")
    assert np.array_equal(a["cell"][ia], b["cell"][ib].astype(np.int32))
    for k in ("x", "ux", "uy", "uz"):
        assert np.max(np.abs(a[k][ia] - b[k][ib]), initial=0.0) < tol, k
    assert np.array_equal(a["w"][ia], b["w"][ib])


def _state(rng, cells, umax, dx, ids_from=0):
    n = len(cells)
    u = umax * rng.standard_normal((3, n))
    w = 1.0e3 * (1.0 + (ids_from + np.arange(n)) * 1e-9)  # unique weights double as identities
    return np.asarray(cells, dtype=np.int32), rng.random(n), u[0], u[1], u[2], w


def test_empty_population_through_every_operator(oa, oracle):
    """n = 0: advance, deposit, the fused push, re-binning, energies and read-back are all no-ops (the reference's
    deposit returns early on an empty slice, yee.rs:557-559; its loops over particles simply do not run)."""
    nx, dx = 64, 1.0e-8
    dt = 0.95 * dx / C
    g = oa.Grid.single(nx, 0.0, dx)
    e = np.zeros(0)
    for species in (oa.ELECTRON, oa.ION):
        p = oa.Population.from_soa(species, e.astype(np.int32), e, e, e, e, e, dx, dt)
        assert len(p) == 0
        p.advance(g, dt)
        g.clear()
        g.deposit(p, dt)
        p.advance_deposit(g, dt)
        p.sort(g)
        assert p.total_kinetic_energy() == 0.0
        s = p.soa()
        assert all(len(v) == 0 for v in s.values())
    g.synchronize()
    g.advance(dt)
    assert all(np.all(g.raw(k) == 0.0) for k in FIELDS + ("jx", "jy", "jz", "rho"))


def test_empty_species_beside_a_full_one(oa, oracle):
    """`ions: npc: 0` (the reference's examples/two_stream.yaml): the paired launch serves one species and an empty one."""
    o = oracle
    nx, dx = 96, 500.0
    dt = 0.95 * dx / C
    rng = np.random.default_rng(5)
    st = _state(rng, np.repeat(np.arange(nx), 50), 0.3, dx)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt)
    e = np.zeros(0)
    pi = oa.Population.from_soa(oa.ION, e.astype(np.int32), e, e, e, e, e, dx, dt)
    _steps(oa, o, g, og, [pe, pi], [ope], dt, 12)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)
    assert len(pi) == 0


@pytest.mark.parametrize("n", [1, 2, 63, 65, 255, 257, 1023, 2049, 4097])
def test_ragged_population_sizes(oa, oracle, n):
    """Sizes that are not a multiple of the wavefront (64), the workgroup (256) or a chunk: the lanes past the end are
    masked, nobody deposits twice and nobody is skipped.  One particle included."""
    o = oracle
    nx, dx = 72, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(100 + n)
    cells = np.sort(rng.integers(0, nx, n))
    st = _state(rng, cells, 2.0, dx)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt, sort_interval=4)
    st_i = _state(rng, cells[::-1].copy()[: max(1, n // 2)], 0.02, dx, ids_from=n)
    pi, opi = _pair(oa, o, oa.ION, st_i, dx, dt, charge=(6.0, 6.0, 12.0), sort_interval=4)
    _steps(oa, o, g, og, [pe, pi], [ope, opi], dt, 10)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)
    _cmp_pops(pi, opi)
    assert abs(pe.total_kinetic_energy() / ope.total_kinetic_energy() - 1.0) < 1e-12


@pytest.mark.parametrize("fused", [True, False])
def test_smallest_subdomain(oa, oracle, fused):
    """8 cells (GridDesign's min_subsize, grid/mod.rs:95), periodic, relativistic particles: every particle wraps around
    the whole domain several times in 40 steps; the tile is wider than the subdomain."""
    o = oracle
    nx, dx = 8, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(8)
    st = _state(rng, np.repeat(np.arange(nx), 40), 4.0, dx)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt, sort_interval=5)
    st_i = _state(rng, np.repeat(np.arange(nx), 40), 4.0, dx, ids_from=1000)
    pi, opi = _pair(oa, o, oa.ION, st_i, dx, dt, charge=(1.0, 1.0, A_POS), sort_interval=5)
    _steps(oa, o, g, og, [pe, pi], [ope, opi], dt, 40, fused=fused)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)
    _cmp_pops(pi, opi)


def test_every_particle_in_one_cell(oa, oracle):
    """100 000 particles in cell 37 of 200, the rest of the domain empty: one cell spans dozens of chunks, every LDS
    accumulator of the tile takes all the traffic, re-binning meets a single non-empty bin."""
    o = oracle
    nx, dx, n = 200, 1.0e-8, 100_000
    dt = 0.95 * dx / C
    rng = np.random.default_rng(37)
    st = _state(rng, np.full(n, 37), 0.8, dx)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt, sort_interval=3)
    pe.sort(g)
    _steps(oa, o, g, og, [pe], [ope], dt, 12)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)
    assert pe.counters()["tile_misses"] >= 0


def test_slots_in_random_order(oa, oracle):
    """A population uploaded in no order at all (every chunk holds cells from everywhere: all of them miss the tile and
    take the slow path until the first re-binning) gives the oracle's numbers."""
    o = oracle
    nx, dx = 512, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(13)
    st = _state(rng, rng.integers(0, nx, 20_000), 1.0, dx)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt)
    _steps(oa, o, g, og, [pe], [ope], dt, 10)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)


def test_particles_on_cell_boundaries(oa, oracle):
    """x = 0 exactly, x = the last double below 1, u = 0 and u that carries the particle exactly onto a boundary region:
    the cell arithmetic of the push (particle/mod.rs floor / wrap) and the flux's sign cases (yee.rs:185-204)."""
    o = oracle
    nx, dx = 64, 1.0e-8
    dt = 0.95 * dx / C
    xs = np.array([0.0, np.nextafter(1.0, 0.0), 0.5, np.nextafter(0.5, 0.0), np.nextafter(0.5, 1.0), 2.0 ** -52, 0.05, 0.95])
    us = np.array([0.0, 1e-12, -1e-12, 0.0526, -0.0526, 30.0, -30.0, 1.0, -1.0])
    X, U = np.meshgrid(xs, us, indexing="ij")
    n = X.size
    cells = np.tile(np.array([0, 1, 31, 62, 63]), n // 5 + 1)[:n]
    order = np.argsort(cells, kind="stable")
    w = 1.0e3 * (1.0 + np.arange(n) * 1e-9)
    z = np.zeros(n)
    st = (cells[order].astype(np.int32), X.ravel()[order], U.ravel()[order], z, z, w)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    pe, ope = _pair(oa, o, oa.ELECTRON, st, dx, dt, sort_interval=0)
    _steps(oa, o, g, og, [pe], [ope], dt, 1)
    a, b = pe.soa(), ope.soa()
    ia, ib = np.argsort(a["w"]), np.argsort(b["w"])
    assert np.array_equal(a["cell"][ia], b["cell"][ib].astype(np.int32))  # nobody lands in a different cell
    assert np.max(np.abs(a["x"][ia] - b["x"][ib])) < 1e-13
    assert np.all((a["x"] >= 0.0) & (a["x"] < 1.0))
    for k in ("jx", "rho"):
        ref = og.raw(k)
        assert np.max(np.abs(g.raw(k) - ref)) <= 1e-12 * np.max(np.abs(ref)), k
    _steps(oa, o, g, og, [pe], [ope], dt, 5)
    _cmp_fields(g, og)
    _cmp_pops(pe, ope)


def test_capacity_limits_are_errors(oa, oracle):
    """Slots are 31-bit per subdomain (a cell index and a slot index share i32 arithmetic): asking for more is an error,
    never a wrap-around; an upload larger than the population was created for is refused whole (arrivals from a
    neighbour, by contrast, make it grow: tests/test_gpu_multirank.py)."""
    o = oracle
    with pytest.raises(oa.OpalError, match="2\\^31"):
        oa.Population(oa.ELECTRON, capacity=2 ** 31)
    nx, dx = 64, 1.0e-8
    dt = 0.95 * dx / C
    rng = np.random.default_rng(3)
    st = _state(rng, np.repeat(np.arange(nx), 100), 0.5, dx)
    p = oa.Population(oa.ELECTRON, capacity=1024)
    with pytest.raises(oa.OpalError, match="exceed the capacity"):
        p.upload(*st, dx, dt)
    assert len(p) == 0
    p = oa.Population(oa.ELECTRON, capacity=6400)  # exactly full
    p.upload(*st, dx, dt)
    assert len(p) == 6400
    op = o.Population.from_soa(o.ELECTRON, *st, dx, dt)
    g, og = oa.Grid.single(nx, 0.0, dx), o.Grid.single(nx, 0.0, dx)
    _steps(oa, o, g, og, [p], [op], dt, 5)
    _cmp_fields(g, og)
    _cmp_pops(p, op)
