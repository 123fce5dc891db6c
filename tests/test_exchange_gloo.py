"""N>1 path on CPU: world_size 2 and 3 over gloo.  The transport of opal_amd.exchange carries the
reference's halo + migration traffic between oracle subdomains living in separate processes; the
result must equal, bit for bit, the same decomposition run inside one process."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _decomp_common import A_POSITRON, C, global_state, split  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(worker, world, args, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


@pytest.mark.parametrize("world,left,right", [(2, 0, 0), (3, 0, 0), (2, 1, 2)])
def test_decomposed_oracle_over_gloo(oracle, tmp_path, world, left, right):
    o = oracle
    nx, npc, steps = 48, 12, 12
    dx = 1.0e-8
    dt = 0.95 * dx / C
    _launch("_gloo_oracle_worker.py", world, [tmp_path, nx, npc, steps, left, right])
    # the same decomposition inside this process
    nxs, offs = o.design_unbalanced(world, nx, dx)
    grids = [o.Grid(r, world, left, right, 0.0, dx, nxs, offs) for r in range(world)]
    se, si = global_state(nx, npc, dx, 5), global_state(nx, npc, dx, 6)
    pes = [o.Population.from_soa(o.ELECTRON, *split(se, nxs, r), dx, dt) for r in range(world)]
    pis = [o.Population.from_soa(o.ION, *split(si, nxs, r), dx, dt) for r in range(world)]
    for p in pis:
        p.set_charge_state(1.0, 1.0, A_POSITRON)
    t = 0.0
    for _ in range(steps):
        assert o.world_step(grids, pes, pis, True, t, dt) == 0
        t += dt
    moved = 0
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "jy", "jz", "rho"):
            assert np.array_equal(got[k], grids[r].raw(k)), (r, k)
        for name, p in (("e", pes[r]), ("i", pis[r])):
            ref = p.soa()
            for k, v in ref.items():
                assert np.array_equal(got[f"{name}_{k}"], v), (r, name, k)
        moved += len(pes[r])
    if left == 0:
        assert moved == nx * npc  # periodic: nobody lost
    else:
        assert moved < nx * npc   # open edges: leavers deleted


def test_one_subdomain_equals_many(oracle):
    """N-subdomain fields == 1-subdomain fields (to rounding: the summation order differs)."""
    o = oracle
    nx, npc, steps = 48, 12, 12
    dx = 1.0e-8
    dt = 0.95 * dx / C
    res = {}
    for world in (1, 3):
        nxs, offs = o.design_unbalanced(world, nx, dx)
        grids = [o.Grid(r, world, 0, 0, 0.0, dx, nxs, offs) for r in range(world)]
        se, si = global_state(nx, npc, dx, 5), global_state(nx, npc, dx, 6)
        pes = [o.Population.from_soa(o.ELECTRON, *split(se, nxs, r), dx, dt) for r in range(world)]
        pis = [o.Population.from_soa(o.ION, *split(si, nxs, r), dx, dt) for r in range(world)]
        for p in pis:
            p.set_charge_state(1.0, 1.0, A_POSITRON)
        t = 0.0
        for _ in range(steps):
            o.world_step(grids, pes, pis, True, t, dt)
            t += dt
        res[world] = np.concatenate([g.raw("Ex")[4:-4] for g in grids])
    scale = np.max(np.abs(res[1]))
    assert scale > 0
    assert np.max(np.abs(res[1] - res[3])) / scale < 1e-10


@pytest.mark.parametrize("world", [2, 3])
def test_output_time_collectives_over_gloo(world):
    """exchange.TorchCollectives (the all-reduce and gather callbacks of opal_sim_set_collectives) on CPU."""
    _launch("_gloo_collectives_worker.py", world, [])
