"""Diagnostic: per-100-step kernel time, re-binning count and tile-miss counter over a long two-stream run."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import opal_amd as oa
import torch
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=int(os.environ.get("NPC", "2441"))), stream=torch.cuda.current_stream().cuda_stream)
oa.prof_enable(1)
for blk in range(int(os.environ.get("BLOCKS", "30"))):
    oa.prof_reset()
    t0 = time.perf_counter()
    sim.step(100); sim.sync()
    el = time.perf_counter() - t0
    n, ms = oa.prof_get("fused"); ns, mss = oa.prof_get("sort")
    c = sim.electrons.counters()
    print(blk, "misses", c["tile_misses"], "since", c["steps_since_sort"], "ms/step %.3f" % (el * 10), "fused avg ms %.3f" % (ms / max(n, 1)), "sorts", ns, "sort ms %.2f" % mss, flush=True)
