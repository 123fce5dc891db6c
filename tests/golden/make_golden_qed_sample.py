#!/usr/bin/env python3
"""Golden vectors for photon_emission::sample / classical_sample ON THE REFERENCE'S OWN TABLES.

What this pins: `sample` (photon_emission.rs:129-203) is code + 73 lookup tables.  The C oracle and the HIP
kernel both read opal_amd/csrc/qed_tables.inc, so a GPU-vs-oracle test cannot see a wrong table.  This script
is the independent leg: a pure-Python restatement of the sampler (written from the Rust, sharing no code or
include with oracle/ or csrc/) evaluated on tables parsed AT RUN TIME from
/root/reference/src/qed/photon_emission_tables.rs and photon_emission.rs (tools/ref_qed_tables.py).  Its
inputs and outputs -- numbers only -- are committed as tests/golden/qed_sample_reference_tables.json; the
oracle must reproduce them bit for bit (same libm, same operation order), the HIP path to 1e-12.

The reference cannot run here (Rust; no toolchain), so these are not outputs of the reference binary: they are
the reference's algorithm restated a second time, on the reference's data.

Run in the build container:  python tests/golden/make_golden_qed_sample.py
Follows: pwmci.rs:14-78 (fit), :96-111 (evaluate), :113-228 (invert); photon_emission.rs:87-121
(from_linear_cdf_table), :129-203 (sample), :264-292 (classical_sample).
"""
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_qed_tables as ref  # noqa: E402

T = ref.load()
BRANCHES = set()


class WouldPanic(Exception):
    """The reference indexes table[i-1] with i == 0 (pwmci.rs:63,152): a measure-zero input; not sampled."""


# ---------------------------------------------------------------- pwmci.rs
def hermite(i, tab):
    """FitParameters::construct: (x0, x1, f0, f1, m0, m1) for the segment between rows i-1 and i."""
    if i == 0:
        raise WouldPanic
    n = len(tab)
    slope = lambda a, b: (tab[b][1] - tab[a][1]) / (tab[b][0] - tab[a][0])
    mid = slope(i - 1, i)
    left = mid if i == 1 else slope(i - 2, i - 1)
    right = mid if (i == n - 1 and i != 1) else slope(i, i + 1)  # the i == 1 arm is tested first (:18)
    m0 = 0.5 * (left + mid) if left * mid > 0.0 else 0.0
    m1 = 0.5 * (mid + right) if mid * right > 0.0 else 0.0
    if mid != 0.0 and m0 / mid > 3.0:
        m0 = 3.0 * mid
    if right != 0.0 and m1 / right > 3.0:
        m1 = 3.0 * right
    return tab[i - 1][0], tab[i][0], tab[i - 1][1], tab[i][1], m0, m1


def hermite_at(p, x):
    x0, x1, f0, f1, m0, m1 = p
    t = (x - x0) / (x1 - x0)
    one_t = 1.0 - t
    h00 = (1.0 + 2.0 * t) * (one_t * one_t)
    h10 = t * (one_t * one_t)
    h01 = (t * t) * (3.0 - 2.0 * t)
    h11 = (t * t) * (t - 1.0)
    return f0 * h00 + f1 * h01 + (x1 - x0) * (m0 * h10 + m1 * h11)


def pw_evaluate(x, tab):
    i = 0
    while i < len(tab) and x > tab[i][0]:
        i += 1
    if i >= len(tab) - 1 and x > tab[-1][0]:
        return None
    return hermite_at(hermite(i, tab), x)


def strictly_inside(x, a, b):
    if a < b:
        return a < x < b
    if b < a:
        return b < x < a
    return True


def pw_invert(f, tab):
    i = 0
    while i < len(tab) and f > tab[i][1]:
        i += 1
    if i >= len(tab) - 1 and f > tab[-1][1]:
        return None
    seg = hermite(i, tab)
    # best, previous, the one before, and the one before that: (x, residual)
    a = (tab[i - 1][0], tab[i - 1][1] - f)
    b = (tab[i][0], tab[i][1] - f)
    if abs(b[1]) < abs(a[1]):
        a, b = b, a
    c, d = b, (0.0, 0.0)
    bisected = False
    for _ in range(32):
        if a[1] != c[1] and b[1] != c[1]:
            r, s, t = a[1] / c[1], a[1] / b[1], b[1] / c[1]
            p = s * (t * (r - t) * (c[0] - a[0]) - (1.0 - r) * (a[0] - b[0]))
            q = (t - 1.0) * (r - 1.0) * (s - 1.0)
            guess = a[0] + p / q
        else:
            guess = a[0] - (a[1] * (a[0] - b[0]) / (a[1] - b[1]))
        reject = (not strictly_inside(guess, 0.25 * (3.0 * b[0] + a[0]), a[0])
                  or (bisected and abs(guess - a[0]) >= 0.5 * abs(a[0] - c[0]))
                  or (not bisected and abs(guess - a[0]) >= 0.5 * abs(c[0] - d[0])))
        if reject:
            bisected, guess = True, 0.5 * (a[0] + b[0])
        else:
            bisected = False
        d, c = c, a
        a = (guess, hermite_at(seg, guess) - f)
        if b[1] * a[1] >= 0.0:
            b, a = a, c
        if abs(b[1]) < abs(a[1]):
            a, b = b, a
        if abs(b[0] - a[0]) / abs(a[0]) < 1.0e-6 or a[1] == 0.0:
            break
    return a[0]


# ---------------------------------------------------------------- photon_emission.rs
def draw_above(global_zero, local_zero, rnd, cdf):
    """from_linear_cdf_table"""
    tab = cdf["table"]
    if local_zero < tab[0][0]:
        r_zero = cdf["coeff"] * math.pow(local_zero - global_zero, cdf["power"])
        BRANCHES.add("y:r_zero power law")
    else:
        r_zero = pw_evaluate(local_zero, tab)
        if r_zero is None:
            BRANCHES.add("y:local_zero beyond table")
            return local_zero
        BRANCHES.add("y:r_zero from table")
    r = r_zero + (1.0 - r_zero) * rnd
    if r <= tab[0][1]:
        y = math.exp((math.log(r) - math.log(cdf["coeff"])) / cdf["power"]) + global_zero
        BRANCHES.add("y:power-law head")
    else:
        y = pw_invert(r, tab)
        if y is None:
            y = local_zero
            BRANCHES.add("y:r beyond table")
        else:
            BRANCHES.add("y:inverted")
    if not y >= local_zero:
        # the reference's assert (:119) fires here; this repo's oracle and kernel clamp instead
        BRANCHES.add("y:assert would fire (clamped)")
        y = local_zero
    return y


def polar_angle(gamma, z):
    cos_theta = (gamma - math.pow(z, 2.0 / 3.0) / (2.0 * gamma)) / math.sqrt(gamma * gamma - 1.0)
    return math.acos(max(min(cos_theta, 1.0), -1.0))


def classical_sample(chi, gamma, r1, r2, r3):
    delta = math.acos((-9.0 + 50.0 * r2 - 25.0 * (r2 * r2)) / 16.0)
    zc = (2.0 + 4.0 * math.cos(delta / 3.0)) / (5.0 * (1.0 - r2))
    z = zc * zc * zc
    tab = T["classical"]
    ln_r = math.log(r1)
    if ln_r < tab[0][1]:
        x = T["CLASSICAL_SMALL_X_COEFF"] * math.pow(r1, 0.6)
        BRANCHES.add("classical:power-law head")
    else:
        ln_x = pw_invert(ln_r, tab)
        if ln_x is None:
            ln_x = tab[-1][0]
            BRANCHES.add("classical:clipped")
        else:
            BRANCHES.add("classical:inverted")
        x = math.exp(ln_x)
    u = 3.0 * chi * x / (2.0 * z)
    return u * gamma, polar_angle(gamma, z), 2.0 * math.pi * r3


def ln_u_of(cdf, ln_r):
    tab = cdf["table"]
    if ln_r <= tab[0][1]:
        BRANCHES.add("u:power-law head")
        return (ln_r - math.log(cdf["coeff"])) / cdf["power"]
    v = pw_invert(ln_r, tab)
    if v is None:
        BRANCHES.add("u:clipped to last row")
        return tab[30][0]
    BRANCHES.add("u:inverted")
    return v


def sample(chi, gamma, r1, r2, r3):
    pos = (math.log(chi) - T["LN_CHI_MIN"]) / T["LN_CHI_STEP"]
    if math.log(chi) <= T["LN_CHI_MIN"]:
        om, th, ph = classical_sample(chi, gamma, r1, r2, r3)
        return om * gamma / (gamma + om), th, ph
    w = pos - math.trunc(pos)
    k = int(math.floor(pos))
    if k >= 40:
        raise NotImplementedError  # unimplemented!() at :144
    ln_r = math.log(r1)
    u = math.exp((1.0 - w) * ln_u_of(T["quantum"][k], ln_r) + w * ln_u_of(T["quantum"][k + 1], ln_r))
    beta = 2.0 * u / (3.0 * chi)
    delta = (1.0 + (1.0 + u) * (1.0 + u)) * math.pow(beta, -2.0 / 3.0) / (1.0 + u)
    pos = (math.log(delta) - T["LN_DELTA_MIN"]) / T["LN_DELTA_STEP"]
    w = pos - math.trunc(pos)
    k = int(math.floor(pos))
    if k >= 30:
        BRANCHES.add("delta > 100")
        y = draw_above(0.0, beta, r2, T["y_inf"])
    else:
        BRANCHES.add("delta tabulated")
        y0 = math.pow(delta, -1.5)
        lo = draw_above(y0, beta, r2, T["y"][k])
        hi = draw_above(y0, beta, r2, T["y"][k + 1])
        y = (1.0 - w) * lo + w * hi
    z = max(y / beta, 1.0)
    return gamma * u / (1.0 + u), polar_angle(gamma, z), 2.0 * math.pi * r3


def main():
    rng = random.Random(20261004)
    cases = []

    def add(kind, chi, gamma, r1, r2, r3):
        f = sample if kind == "sample" else classical_sample
        try:
            out = f(chi, gamma, r1, r2, r3)
        except WouldPanic:
            return
        assert all(math.isfinite(v) for v in out), (kind, chi, gamma, r1, r2, r3, out)
        cases.append({"fn": kind, "in": [v.hex() for v in (chi, gamma, r1, r2, r3)], "out": [v.hex() for v in out]})

    # random sweep: chi log-uniform over the whole tabulated range and below it, gamma 10..1e5
    for _ in range(360):
        chi = 10.0 ** rng.uniform(-3.0, 1.995)
        gamma = 10.0 ** rng.uniform(1.0, 5.0)
        add("sample", chi, gamma, rng.random(), rng.random(), rng.random())
    # the corners: power-law head of the energy cdf, its clipped tail, both ends of the angle cdf
    for chi in (0.0100001, 0.013, 0.1, 1.0, 7.3, 50.0, 99.9):
        for r1 in (1e-12, 1e-6, 0.02, 0.5, 0.999, 1.0 - 1e-9, 1.0 - 1e-14):
            for r2 in (1e-9, 0.3, 0.97, 1.0 - 1e-9):
                add("sample", chi, 2000.0, r1, r2, 0.625)
    # exactly on tabulated chi (weight = 0 up to the truncated LN_CHI_MIN / STEP literals)
    for i in (0, 1, 17, 39):
        add("sample", 0.01 * 10.0 ** (i / 10.0), 500.0, 0.4, 0.6, 0.1)
    # low gamma: cos(theta) leaves [-1, 1] and is clamped
    for g in (1.0001, 1.5, 3.0):
        add("sample", 0.5, g, 0.7, 0.999999, 0.9)
    for _ in range(80):
        add("classical", 10.0 ** rng.uniform(-5.0, -2.0), 10.0 ** rng.uniform(1.0, 4.0), rng.random(), rng.random() * 0.999, rng.random())
    for r1 in (1e-9, 1.3e-3, 1.4e-3, 0.5, 1.0 - 1e-9):
        for r2 in (0.0, 1e-9, 0.5, 0.999):
            add("classical", 0.004, 1000.0, r1, r2, 0.5)
    out = {
        "_about": "inputs/outputs of an independent Python restatement of photon_emission::sample and "
                  "classical_sample evaluated on the reference's tables parsed at run time; IEEE-754 hex floats; "
                  "generator: tests/golden/make_golden_qed_sample.py",
        "branches_covered": sorted(BRANCHES),
        "cases": cases,
    }
    path = os.path.join(HERE, "qed_sample_reference_tables.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print("wrote %s: %d cases; branches: %s" % (path, len(cases), ", ".join(sorted(BRANCHES))))


if __name__ == "__main__":
    main()
