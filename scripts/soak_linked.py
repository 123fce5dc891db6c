"""Soak of the linked step (a lone periodic subdomain made its own neighbour through RCCL): a plasma so thin that its
fields do nothing, so every particle must end up exactly where N ballistic steps put it -- a particle pushed twice, or
not at all, in any one step (the failure mode of shared chunks, races at re-binning, lost or duplicated leavers) is off
by a fifth of a cell.  Particles are matched by their momentum, which nothing changes.
usage: soak_linked.py [steps] [cold|hot]      (environment: NX cells, NPC per cell and species, MODES="linked,peer,in place";
linked = RCCL self-exchange, peer = peer-mapped mailboxes written by the step's own kernels)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import opal_amd as oa
from opal_amd.exchange import PeerLink, RcclLink
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
hot = len(sys.argv) > 2 and sys.argv[2] == "hot"
NX, NPC = int(os.environ.get("NX", "2048")), int(os.environ.get("NPC", "48"))
MODES = os.environ.get("MODES", "linked,peer,in place").split(",")
YAML = f"""---
control:
 dx: 1.0e-8
 nx: {NX}
 xmin: 0.0
 start: 0.0
 end: 1.0e-9
 current_deposition: true
 n_outputs: 2
 balance: false
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: {NPC}
 ne: 1.0e5
 ux: 0.2 * (1.0 + 0.3 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: {"0.01 * nrand" if hot else "0.0"}
 uz: 0.0
 output: x:px
ions:
 name: positron
 npc: {NPC}
 Z: 1.0
 A: me/mp
 ni: 1.0e5
 ux: 0.9 * (1.0 + 0.3 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.0
 uz: 0.0
 output: x:px
"""
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
C = 2.997925e8
bad = 0
for mode in MODES:
    stream = torch.cuda.current_stream().cuda_stream
    link = (RcclLink(0, 1, dev, particle_capacity=4096) if mode == "linked" else
            PeerLink(0, 1, dev, particle_capacity=4096, stream=stream) if mode == "peer" else None)  # peer: mailboxes, no transport
    sim = oa.Simulation(yaml_text=YAML, rank=0, numtasks=1, link=link, stream=torch.cuda.current_stream().cuda_stream)
    if link is not None:
        sim.grid.set_self_exchange(True)
    dt, dx = sim.dt, 1.0e-8
    init = {n: p.soa() for n, p in (("e", sim.electrons), ("i", sim.ions))}
    done = 0
    while done < steps:
        sim.step(min(250, steps - done)); done += min(250, steps - done)
    sim.sync()
    for n, p in (("e", sim.electrons), ("i", sim.ions)):
        a, b = init[n], p.soa()
        assert len(a["ux"]) == len(b["ux"]) == NX * NPC, (mode, n, len(b["ux"]))
        oa_, ob = np.argsort(a["ux"], kind="stable"), np.argsort(b["ux"], kind="stable")
        assert np.array_equal(a["ux"][oa_], b["ux"][ob]), (mode, n, "momenta changed")
        g = np.sqrt(1.0 + a["ux"] ** 2 + a["uy"] ** 2 + a["uz"] ** 2)[oa_]
        per_step = C * a["ux"][oa_] / g * dt / dx
        want = (a["cell"][oa_] + a["x"][oa_] + steps * per_step) % NX
        got = (b["cell"][ob] + b["x"][ob]) % NX
        d = np.abs(got - want); d = np.minimum(d, NX - d)
        print(mode, n, "max |position - ballistic| in cells:", d.max(), "steps since the last re-binning:", p.counters()["steps_since_sort"], flush=True)
        bad += int(d.max() > 1e-6)
    if mode == "peer":
        print("peer link:", link.status(), flush=True)
    del sim, link
dist.destroy_process_group()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
