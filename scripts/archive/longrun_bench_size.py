"""Long run of the 1e8-particle bench workload (in place): sustained rate over thousands of steps, particle count, and
energy bookkeeping block by block (the two-stream instability converts drift energy into field energy; the total
must stay put to the scheme's accuracy)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, opal_amd as oa, torch
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=32768, npc=1526), stream=torch.cuda.current_stream().cuda_stream)
n0 = len(sim.electrons) + len(sim.ions)
f0, ke0, ki0 = sim.energies()
tot0 = f0 + ke0 + ki0
blocks, per = int(os.environ.get("BLOCKS", "8")), int(os.environ.get("PER", "1000"))
for b in range(blocks):
    t0 = time.perf_counter()
    sim.step(per); sim.sync()
    el = time.perf_counter() - t0
    f, ke, ki = sim.energies()
    c = sim.electrons.counters()
    print("block", b, "ms/step %.4f" % (el * 1e3 / per), "updates/s %.3e" % (n0 * per / el), "field/total %.3e" % (f / tot0),
          "total drift %.2e" % ((f + ke + ki - tot0) / tot0), "particles", len(sim.electrons) + len(sim.ions) - n0,
          "since re-bin", c["steps_since_sort"], "misses", c["tile_misses"], flush=True)
