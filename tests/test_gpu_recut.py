"""Cuts that follow the plasma (opal_sim_set_recut; sim.cpp: sim_recut).  The reference partitions the domain once, at
t = 0 (GridDesign::balanced, grid/mod.rs:157-206, main.rs:103-110); a decomposed HIP world that MOVES its cuts while it
runs must still be the reference's world: the same fields in the same global cells, the same particles, whatever subdomain
holds them.  Checked against the STATIC oracle world (the oracle never re-cuts), on global arrays.

Subdomains are threads of one process here (tests/_inproc_world.py); tests/test_gpu_multirank.py has the same through
hipIpc mailboxes between processes.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _decomp_common import A_POSITRON, C  # noqa: E402
from test_gpu_world8 import LASER_PLASMA  # noqa: E402

pytestmark = pytest.mark.gpu

COMPS = ("Ex", "Ey", "Ez", "By", "Bz")
KEYS = ("cell", "x", "ux", "uy", "uz", "w")


def run_recutting_world(world, text, steps, every, threshold, max_shift, capacity=4096, growth=True, force_at=(), fused=True):
    """-> per-subdomain dicts: initial state (static partition), final state, final subdomain sizes, re-cut statistics"""
    import opal_amd as oa
    from _inproc_world import InProcCollectives, InProcLink, grid_dump, run_world, soa_dump
    streams = [oa.stream_create() for _ in range(world)]
    os.environ["OPAL_RECUT_DRIFT_MIN"] = "0.05"  # (read by set_recut: these worlds' fronts are slow, and the drift is what is tested)

    def body(r, w):
        st = streams[r]
        link = InProcLink(w, r, particle_capacity=capacity, stream=st, growth=growth)
        sim = oa.Simulation(yaml_text=text, rank=r, numtasks=world, link=link, stream=st, collectives=InProcCollectives(w, r))
        sim.set_fused(fused)
        out = soa_dump(sim, "init_")
        out["init_size"] = sim.grid.size
        if every:
            sim.set_recut(every, threshold, max_shift)
        n = steps if steps else sim.total_steps
        done = 0
        for at in sorted(force_at):
            sim.step(at - done)
            done = at
            sim.recut_now()
        sim.step(n - done)
        sim.sync()
        out.update(soa_dump(sim))
        out.update(grid_dump(sim.grid))
        out["size"], out["L"] = sim.grid.size, sim.grid.left_bdy_size
        out["energies"] = sim.energies()
        out["steps"], out["t"], out["dt"] = n, sim.t, sim.dt
        out["stats"] = sim.recut_stats(world)
        out["link_growth"] = link.growth()
        del sim, link
        return out

    try:
        return run_world(world, body)
    finally:
        os.environ.pop("OPAL_RECUT_DRIFT_MIN", None)
        for st in streams:
            oa.stream_destroy(st)


def global_rows(parts, sizes, lefts, comp):
    """interior cells of every subdomain, in rank order: the domain's row of `comp`"""
    return np.concatenate([np.asarray(p)[lft:lft + n] for p, n, lft in zip(parts, sizes, lefts)])


def global_particles(dicts, sizes, name, prefix=""):
    """a species' particles of all subdomains with GLOBAL cell indices, ordered by (cell, x)"""
    starts = np.concatenate([[0], np.cumsum(sizes)])
    cols = {k: np.concatenate([d[f"{prefix}{name}_{k}"] for d in dicts]) for k in KEYS}
    cols["cell"] = np.concatenate([d[f"{prefix}{name}_cell"].astype(np.int64) + starts[r] for r, d in enumerate(dicts)])
    order = np.lexsort((cols["x"], cols["cell"]))
    return {k: v[order] for k, v in cols.items()}


def oracle_static_world(o, world, got, sizes0, xmin, dx, bdy, charge_state, steps, t0, laser=(None, None)):
    """the oracle's world on the partition the run STARTED with, from the run's own initial particles"""
    offs = np.concatenate([[0.0], np.cumsum(sizes0)[:-1] * dx])
    nxs = np.asarray(sizes0, dtype=np.uint64)
    grids = [o.Grid(r, world, bdy[0], bdy[1], xmin, dx, nxs, offs) for r in range(world)]
    dt = got[0]["dt"]
    pes = [o.Population.from_soa(o.ELECTRON, *[g[f"init_e_{k}"] for k in KEYS], dx, dt) for g in got]
    pis = [o.Population.from_soa(o.ION, *[g[f"init_i_{k}"] for k in KEYS], dx, dt) for g in got]
    for p in pis:
        p.set_charge_state(*charge_state)
    t = t0
    for _ in range(steps):
        assert o.world_step(grids, pes, pis, True, t, dt, laser_y=laser[0], laser_z=laser[1]) == 0
        t += dt
    return grids, pes, pis, t


def compare_worlds(o, got, grids, pes, pis, sizes0, field_tol, x_tol, particle_by_particle=True):
    sizes1 = [int(g["size"]) for g in got]
    assert sum(sizes1) == sum(sizes0)
    assert got[0]["stats"]["sizes"] == sizes1          # every subdomain knows every size (the grid rows gather by them)
    assert all(g["stats"]["sizes"] == sizes1 for g in got)
    lefts_hip = [int(g["L"]) for g in got]
    lefts_o = [int(g.left_bdy_size) for g in grids]
    for k in COMPS:
        ref = global_rows([g.raw(k) for g in grids], sizes0, lefts_o, k)
        mine = global_rows([g[k] for g in got], sizes1, lefts_hip, k)
        assert mine.shape == ref.shape
        scale = max(np.max(np.abs(ref)), 1e-300)
        assert np.max(np.abs(mine - ref)) <= field_tol * scale, (k, np.max(np.abs(mine - ref)) / scale)
    for name, pops in (("e", pes), ("i", pis)):
        ref = global_particles([{f"{name}_{k}": v for k, v in p.soa().items()} for p in pops], sizes0, name)
        mine = global_particles(got, sizes1, name)
        assert len(mine["x"]) == len(ref["x"]), name                      # nobody lost or doubled on the way
        if not particle_by_particle:
            # (a long relativistic run: round-off has grown to ~1e-7 of a cell and a particle on a cell's edge may sit
            # either side of it -- the populations are compared as distributions over the global cells)
            n = sum(sizes0)
            a, b = np.bincount(mine["cell"], minlength=n), np.bincount(ref["cell"], minlength=n)
            assert np.abs(a - b).sum() <= 2e-4 * len(ref["x"]) + 4, (name, np.abs(a - b).sum())
            continue
        assert np.array_equal(mine["cell"], ref["cell"]), name            # the same particles in the same global cells
        assert np.max(np.abs(mine["x"] - ref["x"])) <= x_tol and np.max(np.abs(mine["ux"] - ref["ux"])) <= x_tol * max(1.0, np.max(np.abs(ref["ux"])))
        assert np.array_equal(mine["w"], ref["w"])
    return sizes1


@pytest.mark.parametrize("world,fused", [(4, True), (8, True), (4, False)])
def test_recutting_laser_plasma_world_is_the_static_world(oracle, tmp_path, world, fused):
    """configs[3]'s physics (hole_boring.yaml's set-up on 1600 cells, `balance: true`): the pulse ploughs the electrons of
    the first subdomains across the cuts; with a check every 10 steps the cuts go after them.  The whole run (1473 steps,
    dozens of re-cuts, edges created for 256 leavers that grow) against the oracle on the initial partition: fields and
    energies 1e-6 as test_gpu_world8's static worlds, the populations as distributions over the global cells."""
    import opal_amd as oa
    o = oracle
    text = LASER_PLASMA.format(end=3.0)
    got = run_recutting_world(world, text, 0, every=10, threshold=1.02, max_shift=24, capacity=256, fused=fused)
    steps = got[0]["steps"]
    st = got[0]["stats"]
    assert st["checks"] == (steps - 1) // 10 and st["recuts"] >= 5, st
    # between the plans the cut the front pushes on drifts with it, a cell at a time in the steps' own pushes (a front is slow:
    # the test lowers the speed from which a cut drifts, run_recutting_world)
    # (the fused step; the reference's operator order re-cuts at once and its cuts stand still between two plans)
    if fused:
            assert st["drift_steps"] >= 20, st  # (the velocity a cut ends the run with may have either sign: the plans correct it)
    else:
        assert st["drift_steps"] == 0
    sizes0 = [int(g["init_size"]) for g in got]
    cfgp = tmp_path / "lp.yaml"
    cfgp.write_text(text)
    cfg = oa.Config(path=cfgp)
    ly, lz = cfg.func("laser", "Ey", ["t", "x"]), cfg.func("laser", "Ez", ["t", "x"])
    dx, xmin = 1.0e-6 / 200, -2.0e-6
    grids, pes, pis, t = oracle_static_world(o, world, got, sizes0, xmin, dx, (o.BDY_LASER, o.BDY_ABSORBING), (6.0, 6.0, 12.0),
                                             steps, -4.0e-6 / C, (ly, lz))
    assert abs(got[0]["t"] - t) <= 1e-12 * abs(t)
    sizes1 = compare_worlds(o, got, grids, pes, pis, sizes0, field_tol=1e-6, x_tol=1e-6, particle_by_particle=False)
    assert sizes1 != sizes0
    # what the re-cuts are for: the static partition has lost its balance, the moving one has kept it
    n_static = np.array([len(pe) + len(pi) for pe, pi in zip(pes, pis)], dtype=float)
    n_moving = np.array([len(g["e_x"]) + len(g["i_x"]) for g in got], dtype=float)
    assert n_moving.sum() == n_static.sum()
    imb_static, imb_moving = n_static.max() / n_static.mean(), n_moving.max() / n_moving.mean()
    assert imb_moving <= 1.15 and imb_moving < imb_static, (imb_static, imb_moving, sizes0, sizes1)
    fe, ofe = sum(g["energies"][0] for g in got), sum(g.em_field_energy() for g in grids)
    assert abs(fe - ofe) / ofe < 1e-6


def test_a_few_recuts_leave_every_bit_of_the_physics_alone(oracle, tmp_path):
    """A short horizon, where round-off has not grown: three forced re-cuts inside 45 steps, fields to 1e-9 of the oracle's
    static world, positions to 1e-9."""
    import opal_amd as oa
    o = oracle
    world, steps = 4, 45
    text = LASER_PLASMA.format(end=3.0).replace("start: -4.0e-6/c", "start: 0.5e-6/c")  # the pulse is on the slab from the start
    got = run_recutting_world(world, text, steps, every=0, threshold=1.0, max_shift=12, capacity=256, force_at=(10, 20, 33))
    sizes0 = [int(g["init_size"]) for g in got]
    assert got[0]["stats"]["checks"] == 3
    cfgp = tmp_path / "lp.yaml"
    cfgp.write_text(text)
    cfg = oa.Config(path=cfgp)
    ly, lz = cfg.func("laser", "Ey", ["t", "x"]), cfg.func("laser", "Ez", ["t", "x"])
    dx, xmin = 1.0e-6 / 200, -2.0e-6
    grids, pes, pis, t = oracle_static_world(o, world, got, sizes0, xmin, dx, (o.BDY_LASER, o.BDY_ABSORBING), (6.0, 6.0, 12.0),
                                             steps, 0.5e-6 / C, (ly, lz))
    compare_worlds(o, got, grids, pes, pis, sizes0, field_tol=1e-9, x_tol=1e-9)


NONUNIFORM = """---
control:
 dx: 1.0e-8
 nx: 768
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
 npc: 12
 ne: 1.0e27 * step(x, 1.0e-6, 4.5e-6)
 ux: 0.2 * (1.0 + 0.05 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.01 * nrand
 uz: 0.01 * nrand
 output: x:px
ions:
 name: positron
 npc: 12
 Z: 1.0
 A: me/mp
 ni: 1.0e27 * step(x, 1.0e-6, 4.5e-6)
 ux: 0.2 * (1.0 + 0.05 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.01 * nrand
 uz: 0.01 * nrand
 output: x:px
"""


@pytest.mark.parametrize("world", [3, 6])
def test_periodic_world_recuts_its_inner_cuts(oracle, world):
    """A periodic ring (no laser section: Internal / Internal, main.rs:95-101) of equal subdomains over a slab that fills
    part of the domain: the inner cuts move onto the slab, the wrap cut (global cell 0, where the rows of {i}_grid.dat
    begin) stays; leavers keep crossing all of them, the wrap included."""
    o = oracle
    steps = 40
    got = run_recutting_world(world, NONUNIFORM, steps, every=4, threshold=1.01, max_shift=32, capacity=2048)
    sizes0 = [int(g["init_size"]) for g in got]
    assert sizes0 == [768 // world] * world
    grids, pes, pis, _ = oracle_static_world(o, world, got, sizes0, 0.0, 1.0e-8, (0, 0), (1.0, 1.0, A_POSITRON), steps, 0.0)
    sizes1 = compare_worlds(o, got, grids, pes, pis, sizes0, field_tol=1e-9, x_tol=1e-9)
    assert sizes1 != sizes0 and got[0]["stats"]["recuts"] >= 3
    n_moving = np.array([len(g["e_x"]) + len(g["i_x"]) for g in got], dtype=float)
    n_static = np.array([len(pe) + len(pi) for pe, pi in zip(pes, pis)], dtype=float)
    assert n_moving.max() / n_moving.mean() < n_static.max() / n_static.mean()


def test_recut_off_is_the_reference_partition(oracle):
    """every = 0 (the default): sizes never change, no check is ever made."""
    got = run_recutting_world(3, NONUNIFORM, 12, every=0, threshold=1.0, max_shift=8)
    assert all(g["size"] == g["init_size"] for g in got)
    assert got[0]["stats"]["checks"] == 0 and got[0]["stats"]["recuts"] == 0 and got[0]["stats"]["imbalance"] == 0.0
