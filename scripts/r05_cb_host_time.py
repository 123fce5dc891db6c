#!/usr/bin/env python3
"""Host time per step of the colliding-beams run (how long opal_sim_step takes to ENQUEUE a step) beside the device time:
the slack the host has.  usage (through gpurun): python3 scripts/r05_cb_host_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import opal_amd as oa  # noqa: E402

oa.set_device(0)
oa.set_elision(True)
sim = oa.Simulation(yaml_text=bench.COLLIDING_BEAMS_YAML.format(npc=bench.CB_NPC))
sim.step(bench.CB_WARM + 300)
sim.sync()
for n in (50, 200, 200):
    t0 = time.perf_counter()
    sim.step(n)
    t1 = time.perf_counter()
    sim.sync()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueued in {1e3 * (t1 - t0):.2f} ms = {1e6 * (t1 - t0) / n:.1f} us per step; done after {1e3 * (t2 - t0):.2f} ms = "
          f"{1e6 * (t2 - t0) / n:.1f} us per step; photons {sim.photons.live()}", flush=True)
