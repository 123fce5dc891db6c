#!/usr/bin/env python3
"""Wall time of setting up BASELINE configs[2] (1e8 particles: the loader's draws on the host, the upload, the first re-binning)
with the loader on one host thread (rounds 1-4) and on the box's threads (round 5: host_setup.cpp load_population).
usage (through gpurun): python3 scripts/r05_loader_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (the workload's input file)
import opal_amd as oa  # noqa: E402
from opal_amd import _lib  # noqa: E402

lib = _lib.load()
oa.set_device(0)
text = bench.YAML.format(nx=bench.C3_CELLS, npc=bench.C3_NPC)
warm = oa.Simulation(yaml_text=bench.YAML.format(nx=256, npc=16))  # (the first HIP calls of a process are not the loader's)
del warm
for threads in (1, 0, 1, 0):
    lib.opal_set_loader_threads(threads)
    t0 = time.perf_counter()
    sim = oa.Simulation(yaml_text=text)
    sim.sync()
    dt = time.perf_counter() - t0
    n = len(sim.electrons) + len(sim.ions)
    print(f"loader threads {'auto (%d CPUs)' % len(os.sched_getaffinity(0)) if threads == 0 else threads}: {n} particles set up in {dt:.2f} s", flush=True)
    del sim
