"""Diagnostic: the linked step (peer self-exchange, one GPU's share of configs[2] at N = 8) with the re-binning policy
adaptive (default) and switched off: what the policy's bookkeeping costs per step, and how long the host takes to
enqueue a step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import opal_amd as oa
import torch
from opal_amd.exchange import PeerLink
stream = torch.cuda.current_stream().cuda_stream
for policy in ("adaptive", "off", "adaptive", "off"):
    link = PeerLink(0, 1, particle_capacity=1024, stream=stream)
    sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=1526), link=link, stream=stream)
    sim.grid.set_self_exchange(True)
    sim.step(300); sim.sync()
    if policy == "off":
        sim.electrons.set_sort_interval(0); sim.ions.set_sort_interval(0)
    sim.step(100); sim.sync()
    t0 = time.perf_counter(); sim.step(2000); host = time.perf_counter() - t0; sim.sync(); tot = time.perf_counter() - t0
    print(f"policy {policy}: {tot / 2000 * 1e6:.1f} us/step (host returned after {host / 2000 * 1e6:.1f} us/step)", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.step(6); host = time.perf_counter() - t0; sim.sync()
    print(f"   enqueue of 6 steps on an idle device: {host / 6 * 1e6:.1f} us/step", flush=True)
    del sim, link
