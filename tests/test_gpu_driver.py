"""The YAML driver end to end on the GPU (config C1's physics: the two-stream example, shortened):
opal_sim_run's outputs against the oracle started from the same initial particles."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C = 2.997925e8

# same physics as the reference's two-stream example (1000 cells of 500 m, 100 electrons per cell,
# ne = 20 m^-3, drift 2.5e-24 kg m/s with 0.1 % spread), run for 4 x 150 steps instead of 20 x 3155
TWO_STREAM = """---
control:
 dx: 500
 nx: 1000
 xmin: 0.0
 start: 0.0
 end: 600.5 * 0.95 * 500 / c
 current_deposition: true
 n_outputs: 4

qed:
 photon_emission: false
 photon_absorption: false

electrons:
 npc: 100
 ne: 20.0
 ux: drift_u * (1.0 + 0.001 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.0
 uz: 0.0
 output: x:px

ions:
 npc: 0

constants:
 drift_u: 2.5e-24 / (me * c)
"""


def _read_energy(path):
    return {k: v for k, v in (line.split() for line in open(path))}


def test_two_stream_yaml_run(oracle, tmp_path):
    import opal_amd as oa
    o = oracle
    yml = tmp_path / "two_stream.yaml"
    yml.write_text(TWO_STREAM)
    sim = oa.Simulation(yaml_path=yml)
    assert sim.total_steps == 600 and sim.steps_between_outputs == 150 and sim.n_outputs == 4
    assert len(sim.electrons) == 100000 and sim.ions is None
    dx, dt = 500.0, sim.dt
    assert dt == 0.95 * dx / C
    init = sim.electrons.soa()
    og = o.Grid.single(1000, 0.0, dx)
    op = o.Population.from_soa(o.ELECTRON, init["cell"], init["x"], init["ux"], init["uy"], init["uz"], init["w"], dx, dt)
    sim.run()
    # the oracle's history, main.rs:217-276
    t = 0.0
    for i in range(5):
        en = _read_energy(tmp_path / f"{i}_energy.dat")
        assert list(en) == ["em_field", "electrons", "ions", "photons"]
        fe, ke = og.em_field_energy(), op.total_kinetic_energy()
        # 7 significant digits are printed (main.rs:35-38): equal strings unless a digit boundary is hit
        assert abs(float(en["electrons"]) - ke) <= 2e-7 * ke
        if fe > 0:
            assert abs(float(en["em_field"]) - fe) <= 2e-7 * fe + 1e-9 * fe
        assert en["ions"] == "0.000000e0" and en["photons"] == "0.000000e0"
        rows = np.loadtxt(tmp_path / f"{i}_grid.dat")
        assert rows.shape == (1000, 11)
        ref = og.interpolate()
        assert np.array_equal(rows[:, 0], ref[:, 0])  # x column
        for col in (2, 5):  # jx, Ex
            scale = max(np.max(np.abs(ref[:, col])), 1e-300)
            assert np.max(np.abs(rows[:, col] - ref[:, col])) / scale < 1e-8, (i, col)
        if i < 4:
            for _ in range(150):
                assert o.world_step([og], [op], None, True, t, dt) == 0
                t += dt
    assert abs(sim.t - t) < 1e-12 * t
    # the `output: x:px` spec of the input file became one FITS image per output (mod.rs:383-568)
    for i in range(5):
        assert (tmp_path / f"{i}_electron_x-px.fits").stat().st_size % 2880 == 0
    # Rust float formatting of the grid dump: no "e+00", shortest round-trip digits
    first = open(tmp_path / "4_grid.dat").readline().split()
    assert first[0] == "0e0" and "e+" not in " ".join(first)


# the physics of the reference's laser-plasma example (laser from the left, absorbing right boundary,
# electrons + carbon ions with thermal momenta), scaled down: 3000 cells, a0 = 30, 20 particles per cell
HOLE_BORING = """---
control:
 dx: micro / 200
 nx: 3000
 xmin: -10*micro
 start: -11.0e-6/c
 end: -4.0e-6/c
 current_deposition: true
 n_outputs: 2
 balance: true

qed:
 photon_emission: false
 photon_absorption: false

electrons:
 npc: 20
 ne: density * critical(omega) * step(x,xmin,xmax)
 ux: sqrt(kT/(m*c^2)) * nrand
 uy: sqrt(kT/(m*c^2)) * nrand
 uz: sqrt(kT/(m*c^2)) * nrand
 output: [x:px, x:p_perp, py:pz]

ions:
 name: carbon
 npc: 20
 Z: Z
 A: A
 ni: density * critical(omega) * step(x,xmin,xmax) / Z
 ux: sqrt(kT/(A*mp*c^2)) * nrand
 uy: sqrt(kT/(A*mp*c^2)) * nrand
 uz: sqrt(kT/(A*mp*c^2)) * nrand
 output: [x:px, x:p_perp, py:pz]

laser:
 Ey: (a0*me*c*omega/e) * gauss_pulse_re(t,x,omega,sigma)
 Ez: (a0*me*c*omega/e) * gauss_pulse_im(t,x,omega,sigma)

constants:
 density: 20.0
 a0: 30.0
 omega: 2*pi*c/0.8e-6
 sigma: pi * 2.0 / sqrt(ln(2.0))
 xmin: -8.0 * micro
 xmax: -6.0 * micro
 Z: 6.0
 A: 12.0
 kT: 500 * eV
"""


def test_laser_plasma_yaml_steps(oracle, tmp_path):
    """Laser + absorbing boundaries, two species, relativistic motion, deletion at open edges, adaptive
    re-binning: fields and energies against the oracle over 900 steps."""
    import opal_amd as oa
    o = oracle
    yml = tmp_path / "hb.yaml"
    yml.write_text(HOLE_BORING)
    sim = oa.Simulation(yaml_path=yml)
    cfg = oa.Config(path=yml)
    ly, lz = cfg.func("laser", "Ey", ["t", "x"]), cfg.func("laser", "Ez", ["t", "x"])
    dx, dt = 1.0e-6 / 200, sim.dt
    g = sim.grid
    assert (g.left_bdy_size, g.right_bdy_size, g.size) == (4, 200, 3000)
    ne, ni = len(sim.electrons), len(sim.ions)
    assert ne == ni == 400 * 20
    ie, ii = sim.electrons.soa(), sim.ions.soa()
    og = o.Grid.single(3000, -10e-6, dx, o.BDY_LASER, o.BDY_ABSORBING)
    ope = o.Population.from_soa(o.ELECTRON, ie["cell"], ie["x"], ie["ux"], ie["uy"], ie["uz"], ie["w"], dx, dt)
    opi = o.Population.from_soa(o.ION, ii["cell"], ii["x"], ii["ux"], ii["uy"], ii["uz"], ii["w"], dx, dt)
    opi.set_charge_state(6.0, 6.0, 12.0)
    t = sim.t
    ope_ke0 = ope.total_kinetic_energy()
    nsteps = 900  # the pulse reaches the plasma after ~630 steps
    for k in range(nsteps):
        assert o.world_step([og], [ope], [opi], True, t, dt, laser_y=ly, laser_z=lz) == 0
        t += dt
    sim.step(nsteps)
    sim.sync()
    assert abs(sim.t - t) <= 1e-12 * abs(t)
    fe, ke, ki = sim.energies()
    ofe, oke, oki = og.em_field_energy(), ope.total_kinetic_energy(), opi.total_kinetic_energy()
    assert ofe > 0 and oke > 0
    # north_star's tolerance: fields within 1e-6 relative of the CPU path
    assert abs(fe - ofe) / ofe < 1e-6
    assert abs(ke - oke) / oke < 1e-6
    assert abs(ki - oki) / oki < 1e-6
    assert oke > 100 * ope_ke0  # the laser has heated the electrons: the interaction really happened
    for comp in ("Ey", "Ez", "By", "Bz", "Ex"):
        a, b = g.raw(comp), og.raw(comp)
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-6, comp
    assert len(sim.electrons) == len(ope) and len(sim.ions) == len(opi)


def test_unfused_driver_equals_fused(tmp_path):
    """opal_sim_set_fused(0): the reference's operator order (advance, advance, clear, deposit, deposit) with both
    charged species in one launch per operator, against the fused pass."""
    import sys
    import opal_amd as oa
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _gloo_hip_worker import YAML
    out = {}
    for fused in (True, False):
        sim = oa.Simulation(yaml_text=YAML.format(nx=96, npc=24), output_dir=str(tmp_path))
        sim.set_fused(fused)
        sim.electrons.set_sort_interval(7)
        sim.ions.set_sort_interval(7)
        sim.step(30)
        sim.sync()
        out[fused] = {k: sim.grid.raw(k) for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "rho")}
        out[fused]["ke"] = sim.energies()
        out[fused]["n"] = (len(sim.electrons), len(sim.ions))
    assert out[True]["n"] == out[False]["n"] == (96 * 24, 96 * 24)
    for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "rho"):
        a, b = out[False][k], out[True][k]
        assert np.max(np.abs(a - b)) <= 1e-9 * max(np.max(np.abs(b)), 1e-300), k
    assert np.allclose(out[False]["ke"], out[True]["ke"], rtol=1e-10)


def test_device_identity():
    """opal_device_identity: the PCI bus id and name of the card a rank runs on (what bench.py's `config.ranks` reports)."""
    import re
    import opal_amd as oa
    bus, name = oa.device_identity()
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", bus), bus
    assert "gfx" in name
    assert oa.device_identity(0) == (bus, name) or oa.device_count() > 1



def test_loader_threads_do_not_change_the_initial_state():
    """A Simulation loads its species with several host threads by default (host_setup.cpp: the one generator's state at any
    cell is known, because every particle takes a fixed number of draws).  The particles -- electrons, then the ions, which
    continue the same generator (main.rs:53, 116-147) -- must be those of the serial loop, bit for bit."""
    import opal_amd as oa
    from opal_amd import _lib
    lib = _lib.load()
    text = TWO_STREAM.replace("nx: 1000", "nx: 1024").replace("npc: 100", "npc: 400")
    text = text.replace("ions:\n npc: 0\n", "ions:\n name: positron\n npc: 300\n Z: 1.0\n A: me/mp\n ni: 20.0\n"
                                             " ux: 0.01 * nrand\n uy: 0.0\n uz: 0.001 * urand\n output: x:px\n")
    assert "positron" in text
    states = []
    for threads in (1, 6):
        assert lib.opal_set_loader_threads(threads) == 0
        try:
            sim = oa.Simulation(yaml_text=text)
        finally:
            lib.opal_set_loader_threads(0)
        states.append((sim.electrons.soa(), sim.ions.soa() if sim.ions is not None else None))
        del sim
    (e1, i1), (e6, i6) = states
    assert len(e1["x"]) == 1024 * 400
    for a, b in ((e1, e6), (i1, i6)):
        if a is None:
            continue
        for k in a:
            assert np.array_equal(a[k], b[k]), k
