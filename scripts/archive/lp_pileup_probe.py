#!/usr/bin/env python3
"""How far does the hole-boring front pile the particles up?  The largest number of same-species particles in one cell,
in units of the initial particles per cell, through the laser-plasma interaction of bench.py's configs[3] physics: what
a link must have room for when the front crosses a cut (leavers per step <= the occupants of the cell next to the cut).
usage: python scripts/lp_pileup_probe.py [npc] [sample every]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opal_amd as oa
import bench

npc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sim = oa.Simulation(yaml_text=bench.LASER_PLASMA_YAML.format(npc=npc))
sim.step(19000)
worst = {}
for s in range(19000, 25001, every):
    for name, p in (("electrons", sim.electrons), ("ions", sim.ions)):
        c = p.soa(fields=())["cell"] if False else p.soa()["cell"]
        m = np.bincount(c[c >= 0]).max() / npc
        worst[name] = max(worst.get(name, 0.0), m)
        if s % 500 == 0:
            print(f"step {s} {name}: densest cell holds {m:.1f} x npc", flush=True)
    sim.step(every)
print("worst", worst)
