#!/usr/bin/env python3
"""Generates tests/golden/qed_radiate_independent.json: Electron::radiate's event, restated a second time.

The sampler has had its independent leg since round 2 (make_golden_qed_sample.py: photon_emission::sample on the
reference's own tables, parsed at run time from the reference's files).  This file adds the rest of the event,
electron.rs:213-240 -- the photon's direction from (theta, phi) about the electron's momentum (vec3.rs:115-144: normalize,
orthogonal, rotate_around), its momentum, the formation length, the recoil (u, gamma, chi) -- and the three filters of
emit_radiation (interactions.rs:73-98), in plain Python floats (`math` only), from the Rust source, sharing nothing with
oracle/ or opal_amd/csrc/.  Inputs: chi, gamma, u and the three uniforms the sampler is handed (the reference draws them
from a generator this repository does not have: the uniforms are inputs here, not outputs of anything).
tests/test_qed_radiate_independent.py: the C oracle must reproduce every number bit for bit (CPU); tests/test_gpu_qed.py:
the device's event to 1e-12.

Runs in the build container only (imports make_golden_qed_sample.py, which parses the reference's tables).
"""
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_qed_sample as smp  # noqa: E402  (the independent sampler, on the reference's tables)

SPEED_OF_LIGHT = 2.997925e8       # constants.rs:4
COMPTON_TIME = 1.28808867e-21     # constants.rs:25
ELECTRON_MASS_MEV = 0.510999      # constants.rs:19


def dot(a, b):  # vec3.rs:56-61
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def normalize(a):  # vec3.rs:117-121
    mag = math.sqrt(dot(a, a))
    assert mag > 0.0
    return (a[0] / mag, a[1] / mag, a[2] / mag)


def orthogonal(a):  # vec3.rs:123-130
    if abs(a[0]) > abs(a[2]):
        perp = (-a[1], a[0], 0.0)
    else:
        perp = (0.0, -a[2], a[1])
    return normalize(perp)


def rotate_around(v, axis, theta):  # vec3.rs:132-144
    s, c = math.sin(theta), math.cos(theta)
    ax, ay, az = axis
    return (
        (c + ax * ax * (1.0 - c)) * v[0] + (ax * ay * (1.0 - c) - az * s) * v[1] + (ax * az * (1.0 - c) + ay * s) * v[2],
        (ay * ax * (1.0 - c) + az * s) * v[0] + (c + ay * ay * (1.0 - c)) * v[1] + (ay * az * (1.0 - c) - ax * s) * v[2],
        (az * ax * (1.0 - c) - ay * s) * v[0] + (az * ay * (1.0 - c) + ax * s) * v[1] + (c + az * az * (1.0 - c)) * v[2],
    )


def radiate(chi, gamma, u, r1, r2, r3):
    """electron.rs:213-240 with default features; returns (k, u', gamma', chi', formation_length)."""
    omega_mc2, theta, cphi = smp.sample(chi, gamma, r1, r2, r3)
    parallel = normalize(u)
    perp = orthogonal(parallel)
    perp = rotate_around(perp, parallel, cphi)
    ct, st = math.cos(theta), math.sin(theta)
    # omega_mc2 * (theta.cos() * parallel + theta.sin() * perp)
    k = tuple(omega_mc2 * (ct * parallel[i] + st * perp[i]) for i in range(3))
    formation_length = 2.0 * (gamma * gamma) * theta * SPEED_OF_LIGHT * COMPTON_TIME / chi
    # self.u = self.u - u  (vec3.rs:48-53)
    un = (u[0] - k[0], u[1] - k[1], u[2] - k[2])
    new_gamma = math.sqrt(1.0 + dot(un, un))
    new_chi = chi * new_gamma / gamma
    return k, un, new_gamma, new_chi, formation_length


def filters(k, formation_length, min_energy, max_angle, max_formation_length):
    """interactions.rs:73-98; None = no limit"""
    energy = math.sqrt(dot(k, k)) * ELECTRON_MASS_MEV  # photon.rs:219-221
    energy_ok = True if min_energy is None else energy >= min_energy
    if max_angle is None:
        angle_ok = True
    else:
        p = (k[0] * ELECTRON_MASS_MEV, k[1] * ELECTRON_MASS_MEV, k[2] * ELECTRON_MASS_MEV)  # photon.rs:209-211
        magnitude = math.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
        angle_ok = math.acos(-p[0] / magnitude) <= max_angle
    fl_ok = True if max_formation_length is None else formation_length < max_formation_length
    return energy_ok, angle_ok, fl_ok


def hx(v):
    return [hx(a) for a in v] if isinstance(v, (tuple, list)) else float(v).hex()


def main():
    rng = random.Random(77)
    cases = []
    branches = set()
    for n in range(240):
        chi = 10.0 ** rng.uniform(-2.0, 1.9)
        gscale = 10.0 ** rng.uniform(1.0, 4.0)
        kind = n % 4
        if kind == 0:      # the colliding-beams geometry: momentum along -x
            u = (-gscale * (1.0 + 0.01 * rng.gauss(0, 1)), 1e-3 * gscale * rng.gauss(0, 1), 1e-3 * gscale * rng.gauss(0, 1))
        elif kind == 1:    # |ux| <= |uz|: the other arm of Vec3::orthogonal
            u = (0.1 * gscale * rng.gauss(0, 1), 0.3 * gscale * rng.gauss(0, 1), gscale)
        else:
            u = tuple(gscale * rng.gauss(0, 1) for _ in range(3))
        branches.add("orthogonal: |x| > |z|" if abs(normalize(u)[0]) > abs(normalize(u)[2]) else "orthogonal: |x| <= |z|")
        gamma = math.sqrt(1.0 + dot(u, u))
        r1, r2, r3 = rng.random(), rng.random(), rng.random()
        try:
            k, un, g2, chi2, fl = radiate(chi, gamma, u, r1, r2, r3)
        except smp.WouldPanic:
            continue
        lim = (10.0 ** rng.uniform(-3, 1), 10.0 ** rng.uniform(-4, 0), 10.0 ** rng.uniform(-9, -5))
        keep = filters(k, fl, *lim)
        for name, ok in zip(("energy", "angle", "formation length"), keep):
            branches.add(f"filter {name}: {'kept' if ok else 'dropped'}")
        cases.append({"chi": hx(chi), "gamma": hx(gamma), "u": hx(u), "r": hx((r1, r2, r3)), "k": hx(k), "u_after": hx(un),
                      "gamma_after": hx(g2), "chi_after": hx(chi2), "formation_length": hx(fl),
                      "limits": hx(lim), "kept_by": [bool(b) for b in keep]})
    out = {"_about": "Electron::radiate (electron.rs:213-240) and emit_radiation's filters (interactions.rs:73-98) restated in "
                     "pure Python on the independent sampler of make_golden_qed_sample.py (the reference's tables); hex floats; "
                     "generator: tests/golden/make_golden_qed_radiate.py",
           "branches_covered": sorted(branches), "cases": cases}
    path = os.path.join(HERE, "qed_radiate_independent.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print("wrote %s: %d cases; branches: %s" % (path, len(cases), "; ".join(sorted(branches))))


if __name__ == "__main__":
    main()
