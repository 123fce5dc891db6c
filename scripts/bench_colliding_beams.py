"""Diagnostic: BASELINE configs[4] (examples/colliding_beams.yaml as shipped: gamma = 1000 beam against an
a0 = 20 four-cycle pulse, current_deposition off, photon emission on, photons kept within 50 mrad) on one
GPU with `npc` scaled up; reports electron- and photon-updates/s before, during and after the collision."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opal_amd as oa
import torch

YAML = """---
control:
 dx: 0.01*micro
 nx: 4000
 xmin: -15*micro
 start: -24.0e-6/c
 end: 6.0e-6/c
 current_deposition: false
 n_outputs: 5
qed:
 photon_emission: true
 photon_absorption: false
 photon_angle_max: 50 * milli
electrons:
 npc: {npc}
 ne: S * a0 * critical(omega) * step(x,xmin,xmax)
 ux: -1000.0 * (1.0 + 0.01 * nrand)
 uy: 0.0
 uz: 0.0
 output: [x, chi, x:chi, energy, x:energy]
ions:
 npc: 0
photons:
 npc: 0
 output: [x, energy, energy:(log;energy), longitude:latitude, longitude:latitude:(energy)]
laser:
 Ey: >
  (a0*m*c*omega/e)
  *sin(omega*(t-x/c))
  *exp(-ln(2.0)*(omega*(t-x/c))^2/(2.0*pi^2*ncycles^2))
 Ez: 0.0
constants:
 S: 1.0e-6
 a0: 20.0
 omega: 2*pi*c/0.8e-6
 ncycles: 4.0
 xmin: (20 - 2.5) * micro
 xmax: (20 + 2.5) * micro
"""
npc = int(os.environ.get("NPC", "20000"))
t0 = time.perf_counter()
sim = oa.Simulation(yaml_text=YAML.format(npc=npc), stream=torch.cuda.current_stream().cuda_stream)
ne = len(sim.electrons)
print("electrons", ne, "total steps", sim.total_steps, "setup s %.1f" % (time.perf_counter() - t0), flush=True)
blk = int(os.environ.get("BLK", "250"))
nblk = int(os.environ.get("NBLK", str(sim.total_steps // blk)))
for b in range(nblk):
    nph0 = sim.photons.counters()["slots"]
    t0 = time.perf_counter()
    sim.step(blk); sim.sync()
    el = time.perf_counter() - t0
    fe, ke, _ = sim.energies()
    nph1 = sim.photons.counters()["slots"]
    nel = sim.electrons.counters()["slots"]
    upd = (nel + 0.5 * (nph0 + nph1)) * blk
    print("block", b, "t=%.3e" % sim.t, "ms/step %.3f" % (el * 1e3 / blk), "electrons %d photons %d" % (nel, nph1),
          "updates/s %.3e" % (upd / el), "KE_e %.4e KE_ph %.4e" % (ke, sim.photons.total_kinetic_energy()), flush=True)
