"""Diagnostic: tile misses per step and kernel time of the laser-plasma workload's quiet phase."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, opal_amd as oa, torch
npc = int(os.environ.get("NPC", "2000"))
sim = oa.Simulation(yaml_text=bench.LASER_PLASMA_YAML.format(npc=npc), stream=torch.cuda.current_stream().cuda_stream)
sim.step(int(os.environ.get("SKIP", "0"))); sim.sync()
for b in range(8):
    c0e, c0i = sim.electrons.counters(), sim.ions.counters()
    t0 = time.perf_counter(); sim.step(5); sim.sync(); el = time.perf_counter() - t0
    ce, ci = sim.electrons.counters(), sim.ions.counters()
    print("ms/step %.3f" % (el * 200), "e: misses %d (+%d) since_sort %d ppt %d slots %d | i: misses %d (+%d) since_sort %d" % (
        ce["tile_misses"], ce["tile_misses"] - c0e["tile_misses"], ce["steps_since_sort"], ce["ppt"], ce["slots"],
        ci["tile_misses"], ci["tile_misses"] - c0i["tile_misses"], ci["steps_since_sort"]), flush=True)
