"""One step of the 1e8-particle two-stream bench workload through the electrostatic launch and through the general
kernel (opal_set_elision(0)): every particle must end up with the same bits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, opal_amd as oa, torch
res = []
for elision in (True, False):
    oa.set_elision(elision)
    sim = oa.Simulation(yaml_text=bench.YAML.format(nx=32768, npc=1526), stream=torch.cuda.current_stream().cuda_stream)
    sim.step(3); sim.sync()
    res.append({n: p.soa(fields=("x", "ux")) for n, p in (("e", sim.electrons), ("i", sim.ions))})
    res[-1]["Ex"] = sim.grid.raw("Ex")
    del sim
oa.set_elision(True)
ok = True
for n in ("e", "i"):
    for f in ("cell", "x", "ux"):
        same = np.array_equal(res[0][n][f], res[1][n][f])
        print(n, f, "identical" if same else "DIFFERENT (max |d| %g)" % np.max(np.abs(res[0][n][f] - res[1][n][f])))
        ok &= same or f != "cell"
d = np.max(np.abs(res[0]["Ex"] - res[1]["Ex"])) / np.max(np.abs(res[1]["Ex"]))
print("Ex relative difference after 3 steps (summation order of the atomics only):", d)
print("OK" if ok and d < 1e-11 else "FAILED")
