"""Diagnostic: which slots miss their workgroup's LDS tile in the sustained two-stream run (BASELINE configs[2] on one
GPU), through opal_pop_debug_misses.  Round 2 used it to find that the only misses of the periodic run were the
particles that had wrapped around (workgroup 0 and the last one) -- see DESIGN.md §6 (modular tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, opal_amd as oa, torch
from opal_amd import _lib
lib = _lib.load()
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=32768, npc=1526), stream=torch.cuda.current_stream().cuda_stream)
e = sim.electrons
sim.step(168); sim.sync()
print("after 168 steps:", e.counters())
for b in range(8):
    lib.opal_pop_debug_misses(e.h, 1, None)
    sim.step(6); sim.sync()
    out = np.zeros(1 + 4 * 4096, dtype=np.int32)
    lib.opal_pop_debug_misses(e.h, 0, out.ctypes.data)
    c = e.counters()
    n = int(out[0]); rec = out[1:1 + 4 * min(n, 4096)].reshape(-1, 4)
    print("block", b, "since_sort", c["steps_since_sort"], "misses recorded", n)
    if n:
        print(" workgroups", np.unique(rec[:, 0])[:10], "slots", rec[:5, 1], "cells", np.unique(rec[:, 2])[:10], "tile base cells", np.unique(rec[:, 3])[:10])
