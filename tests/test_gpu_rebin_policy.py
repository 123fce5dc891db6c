"""Re-binnings decided by the adaptive policy, under the paired pushes of the YAML driver's step, must not show in the
results: a hot two-species plasma (relativistic electrons and positrons: every chunk outruns its LDS tile within ~35 steps,
so the policy re-bins both species several times in 160 steps -- hist, scan, LDS-staged scatter, chunk table), every particle
tagged by its weight, against the oracle's world particle by particle; and the same with re-binnings on a fixed schedule."""
import numpy as np
import pytest

from _decomp_common import A_POSITRON  # (me / mp with the reference's constants, as the input file's `A: me/mp`)

pytestmark = pytest.mark.gpu


YAML = """---
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
 npc: {npc}
 ne: 1.0e27
 ux: 0.0
 uy: 0.0
 uz: 0.0
 output: x:px
ions:
 name: positron
 npc: {npc}
 Z: 1.0
 A: me/mp
 ni: 1.0e27
 ux: 0.0
 uy: 0.0
 uz: 0.0
 output: x:px
"""


def _state(rng, nx, npc, hot):
    n = nx * npc
    cell = np.repeat(np.arange(nx, dtype=np.int32), npc)
    return (cell, rng.random(n), hot * rng.standard_normal(n), 0.3 * rng.standard_normal(n), 0.3 * rng.standard_normal(n),
            1.0e19 * (1.0 + 1e-9 * np.arange(n)))


@pytest.mark.parametrize("adaptive", [True, False])
def test_rebinnings_do_not_show_in_the_results(oracle, adaptive):
    import opal_amd as oa
    o = oracle
    nx, npc, steps = 512, 48, 160
    dx = 1.0e-8
    sim = oa.Simulation(yaml_text=YAML.format(nx=nx, npc=npc))
    dt = sim.dt
    rng = np.random.default_rng(21)
    st_e, st_i = _state(rng, nx, npc, 3.0), _state(rng, nx, npc, 2.0)
    sim.electrons.upload(*st_e, dx, dt)
    sim.ions.upload(*st_i, dx, dt)
    if not adaptive:
        sim.electrons.set_sort_interval(30)
        sim.ions.set_sort_interval(45)
    sim.step(steps)
    sim.sync()
    ce, ci = sim.electrons.counters(), sim.ions.counters()
    if adaptive:  # the policy has re-binned both (nobody else does), and not at the very end
        assert 0 < ce["steps_since_sort"] < steps - 30 and 0 < ci["steps_since_sort"] < steps - 30, (ce, ci)
    else:
        assert ce["steps_since_sort"] == steps % 30 and ci["steps_since_sort"] == steps % 45, (ce, ci)
    nxs, offs = o.design_unbalanced(1, nx, dx)
    og = o.Grid(0, 1, 0, 0, 0.0, dx, nxs, offs)
    pe = o.Population.from_soa(o.ELECTRON, *st_e, dx, dt)
    pi = o.Population.from_soa(o.ION, *st_i, dx, dt)
    pi.set_charge_state(1.0, 1.0, A_POSITRON)
    t = 0.0
    for _ in range(steps):
        assert o.world_step([og], [pe], [pi], True, t, dt) == 0
        t += dt
    for comp in ("Ex", "Ey", "Ez", "By", "Bz"):
        a, b = sim.grid.raw(comp), og.raw(comp)
        assert np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300) < 1e-8, comp
    for got, ref in ((sim.electrons.soa(), pe.soa()), (sim.ions.soa(), pi.soa())):
        assert len(got["w"]) == len(ref["w"]) == nx * npc
        a, b = np.argsort(got["w"], kind="stable"), np.argsort(ref["w"], kind="stable")
        assert np.array_equal(got["w"][a], ref["w"][b])
        assert np.array_equal(got["cell"][a], ref["cell"][b].astype(np.int32))
        for k in ("x", "ux", "uy", "uz"):
            assert np.max(np.abs(got[k][a] - ref[k][b])) < 1e-8 * max(1.0, np.max(np.abs(ref[k]))), k
