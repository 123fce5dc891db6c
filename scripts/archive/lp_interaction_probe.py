"""Diagnostic: the laser-plasma interaction step by step -- push kernel time (HIP events around every launch), tile misses
and steps since the electrons' last re-binning: how the push degrades between re-binnings and what a re-binning costs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, opal_amd as oa, torch
npc = int(os.environ.get("NPC", "2000"))
skip = int(os.environ.get("SKIP", "21500"))
if os.environ.get("ELISION", "1") == "0":
    oa.set_elision(False)
sim = oa.Simulation(yaml_text=bench.LASER_PLASMA_YAML.format(npc=npc), stream=torch.cuda.current_stream().cuda_stream)
sim.step(skip); sim.sync()
fixed = int(os.environ.get("SORT_EVERY", "0"))
if fixed:
    sim.electrons.set_sort_interval(fixed); sim.ions.set_sort_interval(fixed)
rows = []
for b in range(int(os.environ.get("N", "100"))):
    c0 = sim.electrons.counters()
    oa.prof_reset(); oa.prof_enable(1)
    t0 = time.perf_counter(); sim.step(1); sim.sync(); el = time.perf_counter() - t0
    oa.prof_enable(0)
    n, ms = oa.prof_get("fused"); ns, sms = oa.prof_get("sort")
    c = sim.electrons.counters(); ci = sim.ions.counters()
    rows.append((el * 1e3, ms, sms, c["tile_misses"] - c0["tile_misses"] if c["steps_since_sort"] else c["tile_misses"], c["steps_since_sort"], ci["steps_since_sort"], c["slots"]))
print("step  wall_ms  push_ms  sort_ms  e_misses(+)  e_since_sort  i_since_sort  e_slots   (wall: one synchronised step, launch latency included)")
for i, r in enumerate(rows):
    print("%4d  %.4f  %.4f  %.4f  %8d  %4d  %4d  %d" % ((i,) + r))
rows = [r[1:] for r in rows]
tot = sum(r[0] + r[1] for r in rows)
print("mean push %.4f ms, sorts %.4f ms/step, n_e+n_i = %d -> %.3e updates/s (kernels only)" % (
    sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows), len(sim.electrons) + len(sim.ions),
    (len(sim.electrons) + len(sim.ions)) * len(rows) / (tot * 1e-3)))
