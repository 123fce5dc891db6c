import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, opal_amd as oa
from test_gpu_fullsize import TWO_STREAM
out = []
for withdraw in (True, False):
    sim = oa.Simulation(yaml_text=TWO_STREAM.format(nx=256, npc=256))
    g = sim.grid
    g.set_raw("By", 2e-6 * np.sin(np.arange(g.len) * (2.0 * np.pi * 3.0 / 256)))
    if withdraw: g.assume_b_zero()
    sim.step(1); sim.sync()
    out.append((sim.electrons.soa(), g.raw("jz"), g.raw("Ez"), sim.electrons.counters()))
(a, ja, ea, ca), (b, jb, eb, cb) = out
print(ca, cb)
for f in ("x", "ux", "uy", "uz"):
    d = np.abs(a[f] - b[f]); i = int(np.argmax(d))
    print(f, d.max(), "at slot", i, "cell", a["cell"][i], b["cell"][i], a[f][i], b[f][i])
print("cells equal", np.array_equal(a["cell"], b["cell"]))
print("jz diff", np.abs(ja - jb).max(), np.abs(jb).max(), "Ez diff", np.abs(ea - eb).max(), np.abs(eb).max())
