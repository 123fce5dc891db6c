"""Diagnostic: how long the host takes to enqueue a step (the call returns before the GPU has run it) against
how long the GPU takes to execute it: the margin by which the step loop is device-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import opal_amd as oa
import torch
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=2441), stream=torch.cuda.current_stream().cuda_stream)
sim.step(50); sim.sync()
for n in (4, 8, 8, 8):
    t0 = time.perf_counter()
    sim.step(n)
    host = time.perf_counter() - t0
    sim.sync()
    tot = time.perf_counter() - t0
    print("steps %d  host enqueue us/step %.1f   executed us/step %.1f" % (n, host / n * 1e6, tot / n * 1e6), flush=True)
