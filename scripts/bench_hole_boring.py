"""Diagnostic: laser-plasma (BASELINE configs[3] physics: a0 = 300 laser on an overdense electron-carbon
slab, laser + absorbing boundaries) on one GPU at reduced particle count; reports particle-updates/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as oa
import torch

YAML = """---
control:
 dx: micro / 1000
 nx: 20000
 xmin: -10*micro
 start: -20.0e-6/c
 end: 10.0e-6/c
 current_deposition: true
 n_outputs: 30
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: {npc}
 ne: density * critical(omega) * step(x,xmin,xmax)
 ux: sqrt(kT/(m*c^2)) * nrand
 uy: sqrt(kT/(m*c^2)) * nrand
 uz: sqrt(kT/(m*c^2)) * nrand
 output: [x:px]
ions:
 name: carbon
 npc: {npc}
 Z: Z
 A: A
 ni: density * critical(omega) * step(x,xmin,xmax) / Z
 ux: sqrt(kT/(A*mp*c^2)) * nrand
 uy: sqrt(kT/(A*mp*c^2)) * nrand
 uz: sqrt(kT/(A*mp*c^2)) * nrand
 output: [x:px]
laser:
 Ey: (a0*me*c*omega/e) * gauss_pulse_re(t,x,omega,sigma)
 Ez: (a0*me*c*omega/e) * gauss_pulse_im(t,x,omega,sigma)
constants:
 density: 100.0
 a0: 300.0
 omega: 2*pi*c/0.8e-6
 sigma: pi * 4.0 / sqrt(ln(2.0))
 xmin: 0.0 * micro
 xmax: 5.0 * micro
 Z: 6.0
 A: 12.0
 kT: 500 * eV
"""
npc = int(os.environ.get("NPC", "1000"))
t0 = time.perf_counter()
sim = oa.Simulation(yaml_text=YAML.format(npc=npc), stream=torch.cuda.current_stream().cuda_stream)
n = len(sim.electrons) + len(sim.ions)
print("particles", n, "total steps", sim.total_steps, "setup s %.1f" % (time.perf_counter() - t0), flush=True)
# the pulse reaches the slab after ~20 um / (0.95 nm) = 21000 steps: jump there by stepping in blocks
blk = int(os.environ.get("BLK", "3000"))
oa.prof_enable(int(os.environ.get("PROF", "0")))
for b in range(int(os.environ.get("NBLK", "9"))):
    oa.prof_reset()
    t0 = time.perf_counter()
    sim.step(blk); sim.sync()
    el = time.perf_counter() - t0
    fe, ke, ki = sim.energies()
    ce, ci = sim.electrons.counters(), sim.ions.counters()
    print("block", b, "t=%.3e" % sim.t, "ms/step %.3f" % (el * 1e3 / blk), "updates/s %.3e" % (n * blk / el),
          "E_field %.3e KE_e %.3e KE_i %.3e" % (fe, ke, ki), "live_e", len(sim.electrons), "ppt", ce["ppt"], "nonempty cells e/i", ce["nonempty_cells"], ci["nonempty_cells"],
          "misses since re-bin e/i", ce["tile_misses"], ci["tile_misses"], "steps since re-bin e/i", ce["steps_since_sort"], ci["steps_since_sort"],
          "fused (n, ms)", oa.prof_get("fused"), "sort (n, ms)", oa.prof_get("sort"), flush=True)
