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
    # Rust float formatting of the grid dump: no "e+00", shortest round-trip digits
    first = open(tmp_path / "4_grid.dat").readline().split()
    assert first[0] == "0e0" and "e+" not in " ".join(first)
