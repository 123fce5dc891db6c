"""Times one re-binning (opal_pop_sort) of a population disordered as the laser-plasma interaction disorders its electrons:
1e7 particles, 2000 per cell in 5000 of 20000 cells, every particle up to DRIFT cells away from the cell its slot was
sorted for.  Prints the KERNEL_SORT time (HIP events round the whole re-binning); run under `rocprofv3 --kernel-trace --stats`
for the kernels one by one.  Usage: python3 scripts/r04_sort_probe.py [reps] [drift]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opal_amd as oa

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
drift = int(sys.argv[2]) if len(sys.argv) > 2 else 13
nx, npc, first = 20000, 2000, 10000
dx = 1.0e-8
dt = 0.95 * dx / 299792458.0
rng = np.random.default_rng(5)
n = 5000 * npc
home = np.repeat(np.arange(first, first + 5000, dtype=np.int32), npc)
g = oa.Grid.single(nx, 0.0, dx)
f = [rng.random(n) for _ in range(4)]
w = np.full(n, 1.0e3)
oa.prof_enable(1)
p = oa.Population(oa.ELECTRON, n + n // 8)
ms = []
for r in range(reps):
    cell = (home + rng.integers(-drift, drift + 1, n)).astype(np.int32)
    p.upload(cell, f[0], f[1], f[2], f[3], w, dx, dt)
    oa.prof_reset()
    p.sort(g)
    launches, total = oa.prof_get("sort")
    ms.append(total / max(launches, 1))
    got = p.soa(fields=("x",))
    c = got["cell"]
    assert len(c) == n and np.all(np.diff(c) >= 0), "not sorted"
    if r == 0:
        order = np.argsort(cell, kind="stable")
        # the same multiset of (cell, x) pairs: within a cell the order is free
        a = np.lexsort((got["x"], c))
        b = np.lexsort((f[0][order], cell[order]))
        assert np.array_equal(c[a], cell[order][b]) and np.array_equal(got["x"][a], f[0][order][b]), "payload mismatch"
print("sort_ms", " ".join(f"{v:.4f}" for v in ms), " median %.4f" % float(np.median(ms)), flush=True)
