"""Which population of the laser-plasma interaction is re-binned how often, and what its misses look like: the interaction
window of bench.py's laser_plasma workload (npc 2000), stepped one step at a time with the counters read after each.
Reading them drains the stream, so the run is slow and its timings mean nothing; the policy's decisions are what is printed."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import opal_amd as oa

npc = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
oa.set_elision(False)
sim = oa.Simulation(yaml_text=bench.LASER_PLASMA_YAML.format(npc=npc))
sim.step(21000)
sim.sync()
import ctypes
import numpy as np
lib = oa._lib.load()
rec = {"e": False, "i": False}
shown = {"e": 0, "i": 0}
last = {"e": 0, "i": 0}
log = {"e": [], "i": []}
for s in range(steps):
    sim.step(1)
    for name, p in (("e", sim.electrons), ("i", sim.ions)):
        c = p.counters()
        if c["steps_since_sort"] < last[name] or (c["steps_since_sort"] == 0 and last[name] == 0 and s):
            log[name].append((s, last[name] + 1, prev[name]["tile_misses"], c["slots"]))
        last[name] = c["steps_since_sort"]
    prev = {"e": sim.electrons.counters(), "i": sim.ions.counters()}
    for name, p in (("e", sim.electrons), ("i", sim.ions)):
        c = prev[name]
        if not rec[name] and c["tile_misses"] == 0 and c["steps_since_sort"] >= (14 if name == "e" else 40) and shown[name] < 3:
            lib.opal_pop_debug_misses(p.h, 1, None)
            rec[name] = True
        elif rec[name] and (c["tile_misses"] > 0 or c["steps_since_sort"] == 0):
            out = np.zeros(1 + 4 * 4096, dtype=np.int32)
            lib.opal_pop_debug_misses(p.h, 0, out.ctypes.data_as(ctypes.c_void_p))
            rec[name] = False
            k = min(int(out[0]), 4096)
            if k:
                shown[name] += 1
                r = out[1:1 + 4 * k].reshape(k, 4)
                off = r[:, 2] - r[:, 3]
                cells = sim.electrons.soa(fields=())["cell"] if name == "e" else sim.ions.soa(fields=())["cell"]
                print(name, "step", s, "since sort", c["steps_since_sort"], "misses recorded", int(out[0]), "cell - tile base: min", off.min(), "max", off.max(),
                      "| cells of the missing", np.unique(r[:, 2])[:12], "| workgroups", np.unique(r[:, 0])[:12], flush=True)
                for wg in np.unique(r[:, 0])[:3]:
                    rows = r[r[:, 0] == wg]
                    lo = int(rows[:, 1].min()) // 2048 * 2048
                    seg = cells[lo:lo + 2048]
                    seg = seg[seg >= 0]
                    print("   workgroup", wg, "tile base", rows[0, 3], "slots near", lo, "hold cells", seg.min() if len(seg) else None, "..", seg.max() if len(seg) else None,
                          "first 256:", seg[:256].min() if len(seg) else None, "..", seg[:256].max() if len(seg) else None, "misses", len(rows), flush=True)
for name in ("e", "i"):
    iv = [l[1] for l in log[name]]
    print(name, "re-binnings", len(log[name]), "in", steps, "steps; intervals", iv[:40], flush=True)
    print(name, "misses counted just before each", [l[2] for l in log[name]][:40], "slots", log[name][-1][3] if log[name] else None, flush=True)
