"""Diagnostic: sustained throughput over many steps (no kernel timers), incl. re-binning."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import opal_amd as oa
import torch
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=2441), stream=torch.cuda.current_stream().cuda_stream)
sim.step(50); sim.sync()
n = int(os.environ.get("STEPS", "1500"))
t0 = time.perf_counter()
sim.step(n); sim.sync()
el = time.perf_counter() - t0
print("steps", n, "ms/step %.4f" % (el * 1e3 / n), "updates/s %.3e" % (2 * 9998336 * n / el), flush=True)
