"""Diagnostic: wall time per step with and without the HIP-event kernel timers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import opal_amd as oa
import torch
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=2441), stream=torch.cuda.current_stream().cuda_stream)
sim.step(20); sim.sync()
for prof in (False, True, False, True):
    oa.prof_reset(); oa.prof_enable(5 if prof else 0)
    t0 = time.perf_counter()
    sim.step(200); sim.sync()
    el = time.perf_counter() - t0
    print("prof", prof, "ms/step %.4f" % (el * 5), "updates/s %.3e" % (2 * 9998336 * 200 / el), flush=True)
