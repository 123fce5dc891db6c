#!/usr/bin/env python3
"""Generates tests/golden/pic_step_vectors.json — a REGRESSION FREEZE of the oracle, not a parity pin.

(The vectors below are the oracle's own output: tests that compare the oracle with this file only show that the oracle has
not drifted since the file was written.  The parity pin of the hot path is tests/golden/pic_step_vectors_independent.json,
written by a second restatement that shares nothing with the oracle: make_golden_pic_independent.py.)

The reference is Rust and cannot be built or run here (no cargo/rustc in the image), so no vector
can be captured from it.  These vectors are produced by the CPU oracle (oracle/opal_oracle.c), which
is itself pinned by the reference's analytic tests (tests/test_oracle_kat.py).  They are therefore
"restatement-generated" fixtures: they freeze the oracle's arithmetic (so that a later edit of the
oracle cannot drift unnoticed) and give the GPU tests fixed inputs/outputs that do not depend on the
oracle being importable.  Every value is written with 17 significant digits.

Contents
  weight / flux : Cell::weight (yee.rs:140-149) and Cell::flux (yee.rs:185-204) on a grid of arguments
                  covering every branch
  electron_push / ion_push : single-particle Vay (electron.rs:268-330) and Boris (ion.rs:168-214) steps
  deposit       : single-particle stencils at x in {0.1, 0.5, 0.9} with and without a cell crossing
                  (yee.rs:584-609, incl. the jy/jz fourth tap that is zero and the rho tap at index-2)
  field_step    : synchronize + advance of a random periodic grid (yee.rs:345-349, 351-452, 839-867)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

C = 2.997925e8


def main():
    o.build()
    out = {"_generator": "tests/golden/make_golden.py (CPU oracle; the Rust reference cannot run here)"}
    args = [-2.5, -1.5, -1.2, -1.0, -0.75, -0.5, -0.25, 0.0, 0.1, 0.5, 0.6, 0.99, 1.0, 1.3, 1.5, 1.6, 2.0]
    out["weight"] = [[a, o.weight(a)] for a in args]
    fl = []
    for xi in [-1.3, -1.0, -0.9, -0.4, -0.1, 0.0, 0.1, 0.45, 0.9, 1.0, 1.2]:
        for d in [-0.9, -0.5, -0.05, 0.0, 0.05, 0.5, 0.9]:
            fl.append([xi, xi - d, o.flux(xi, xi - d)])
    out["flux"] = fl

    rng = np.random.default_rng(2024)
    dx = 1.0e-8
    dt = 0.95 * dx / C
    ep, ip = [], []
    for k in range(12):
        u = (10.0 ** rng.uniform(-3, 2)) * rng.standard_normal(3)
        E = 1e12 * rng.standard_normal(3)
        B = 3e3 * rng.standard_normal(3)
        x = float(rng.random())
        e = o.electron_create(5, x, u, 1.0, dx, dt)
        o.electron_push(e, E, B, dx, dt)
        ep.append({"x": x, "u": list(u), "E": list(E), "B": list(B), "dx": dx, "dt": dt,
                   "out": {"cell": e.cell, "x": e.x, "prev_x": e.prev_x, "u": list(e.u), "gamma": e.gamma,
                           "chi": e.chi, "work": e.work, "y": e.y, "z": e.z}})
        zs, am = (6.0, 12.0) if k % 2 else (1.0, 9.109383e-31 / 1.672622e-27)
        uu = u * (1e-2 if k % 2 else 1.0)
        p = o.ion_create(5, x, uu, 1.0, dx, dt)
        p.Z_star, p.Z, p.A = zs, zs, am
        o.ion_push(p, E, B, dx, dt)
        ip.append({"x": x, "u": list(uu), "E": list(E), "B": list(B), "dx": dx, "dt": dt, "Z": zs, "A": am,
                   "out": {"cell": p.cell, "x": p.x, "prev_x": p.prev_x, "u": list(p.u), "gamma_m1": p.gamma_m1}})
    out["electron_push"] = ep
    out["ion_push"] = ip

    dep = []
    nx = 16
    for x0 in (0.1, 0.5, 0.9):
        for ux in (-0.9, -0.05, 0.0, 0.05, 0.9):  # |u| ~ 1: crosses a cell when x0 is near that edge
            g = o.Grid.single(nx, 0.0, dx)
            p = o.Population.from_soa(o.ELECTRON, [8], [x0], [ux], [0.3], [-0.2], [1.0e6], dx, dt)
            p.advance(g, dt)
            s = p.soa()
            g.clear()
            g.deposit(p, dt)
            dep.append({"x0": x0, "u": [ux, 0.3, -0.2], "w": 1.0e6, "dx": dx, "dt": dt, "nx": nx,
                        "cell": int(s["cell"][0]), "x": float(s["x"][0]), "prev_x": float(s["prev_x"][0]),
                        "jx": g.raw("jx").tolist(), "jy": g.raw("jy").tolist(), "jz": g.raw("jz").tolist(),
                        "rho": g.raw("rho").tolist()})
    out["deposit"] = dep

    nx = 24
    g = o.Grid.single(nx, 0.0, dx)
    init = {}
    for comp in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho"):
        amp = {"E": 1e3, "B": 1e-5, "j": 1e2, "r": 1e-9}[comp[0]]
        a = amp * rng.standard_normal(g.len)
        g.set_raw(comp, a)
        init[comp] = a.tolist()
    g.synchronize(None, None, 0.0, dt)
    g.advance(dt)
    out["field_step"] = {"nx": nx, "dx": dx, "dt": dt, "init": init,
                         "out": {c: g.raw(c).tolist() for c in init}, "energy": g.em_field_energy()}

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pic_step_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
