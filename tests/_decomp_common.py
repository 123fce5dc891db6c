"""Shared set-up for the multi-subdomain tests: a deterministic global two-stream state and its split
into subdomains (what GridDesign::unbalanced + Population::new would give each rank)."""
import numpy as np

C = 2.997925e8
A_POSITRON = 9.109383e-31 / 1.672622e-27


def global_state(nx, npc, dx, seed, thermal=0.05):
    rng = np.random.default_rng(seed)
    n = nx * npc
    ud = 0.2  # fast beams so that many particles cross subdomain edges within a few steps
    cell = np.repeat(np.arange(nx, dtype=np.int64), npc)
    x = rng.random(n)
    ux = ud * (1.0 + thermal * rng.standard_normal(n)) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    uy = 0.01 * rng.standard_normal(n)
    uz = 0.01 * rng.standard_normal(n)
    w = 20.0 * dx / npc * (1.0 + 1e-9 * np.arange(n))  # unique weights = particle identities
    return cell, x, ux, uy, uz, w


def split(state, nxs, r):
    cell, x, ux, uy, uz, w = state
    lo = int(sum(nxs[:r]))
    hi = lo + int(nxs[r])
    m = (cell >= lo) & (cell < hi)
    return cell[m] - lo, x[m], ux[m], uy[m], uz[m], w[m]
