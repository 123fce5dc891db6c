"""Diagnostic: who misses the LDS tile in the linked step (self-exchange) at one GPU's share of configs[2] at N = 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, opal_amd as oa, torch
import torch.distributed as dist
from opal_amd import _lib
from opal_amd.exchange import RcclLink
lib = _lib.load()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
link = RcclLink(0, 1, torch.device("cuda", 0), particle_capacity=1024)
sim = oa.Simulation(yaml_text=bench.YAML.format(nx=4096, npc=1526), link=link, stream=torch.cuda.current_stream().cuda_stream)
sim.grid.set_self_exchange(True)
e = sim.electrons
sim.step(40); sim.sync()
print("after 40 steps:", e.counters())
for b in range(10):
    lib.opal_pop_debug_misses(e.h, 1, None)
    sim.step(4); sim.sync()
    out = np.zeros(1 + 4 * 4096, dtype=np.int32)
    lib.opal_pop_debug_misses(e.h, 0, out.ctypes.data)
    c = e.counters()
    n = int(out[0]); rec = out[1:1 + 4 * min(n, 4096)].reshape(-1, 4)
    wg, cnt = np.unique(rec[:, 0], return_counts=True)
    print("block", b, "since_sort", c["steps_since_sort"], "slots", c["slots"], "misses in 4 steps", n, "workgroups", dict(zip(wg.tolist()[:8], cnt.tolist()[:8])),
          "cells", np.unique(rec[:, 2])[:6], np.unique(rec[:, 2])[-6:], "tile bases", np.unique(rec[:, 3])[:6])
dist.destroy_process_group()
