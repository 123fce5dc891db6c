#!/usr/bin/env python3
"""Generates opal_amd/csrc/qed_tables.inc: lookup tables for sampling the angularly resolved quantum
synchrotron spectrum (the role of the reference's src/qed/photon_emission_tables.rs).

The tables are computed here from the physics, not transcribed: with u = omega / (gamma m - omega),
beta = 2u / (3 chi) and z = [2 gamma^2 (1 - beta_e cos(theta))]^(3/2) >= 1 the double-differential
emission rate is

    d^2 N / (du dz)  ~  beta / (1+u)^3 * [ (1 + (1+u)^2) z^(2/3) - (1+u) ] * K_{1/3}(beta z).

  QUANTUM_CDF[i]   ln cdf(u; chi_i) against ln u, the marginal of the above over z in [1, inf),
                   chi_i = 0.01 * 10^(i/10), i = 0..40; below the first point cdf = coeff * u^(1/3)
  Y_CDF[j]         cdf of y = beta z for fixed delta_j = (1 + (1+u)^2) beta^(-2/3) / (1+u)
                   = 0.1 * 10^(j/10), j = 0..30:  pdf ~ (delta y^(2/3) - 1) K_{1/3}(y) on
                   [delta^(-3/2), inf); below the first point cdf = coeff * (y - delta^(-3/2))^2
  Y_INFINITE_DELTA the delta -> inf limit, pdf ~ y^(2/3) K_{1/3}(y) on [0, inf), cdf = coeff * y^(4/3)
  CLASSICAL_SPECTRUM   ln cdf(x) against ln x (0.02 <= x <= 20) for the classical conditional spectrum,
                   cdf(x) = (3/pi) int_0^x t K_{1/3}(t) dt  (x = 2 u z / (3 chi))

Each table has 31 (41 for the classical one) rows chosen here; the sampling code only needs them to be
monotone.  tests/test_qed_tables.py checks every table against direct quadrature.
Run:  python tools/make_qed_tables.py   (needs scipy; takes about a minute)
"""
import os
import sys

import numpy as np
from scipy.integrate import quad
from scipy.special import kv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 31


def cumulative(f, pts, lo):
    """int_lo^p f for every p in the increasing list pts (piecewise quadrature)."""
    out, acc, a = [], 0.0, lo
    for p in pts:
        acc += quad(f, a, p, limit=200, epsabs=0, epsrel=1e-11)[0]
        out.append(acc)
        a = p
    return np.array(out)


def marginal(u, chi):
    beta = 2.0 * u / (3.0 * chi)
    a, b = 1.0 + (1.0 + u) ** 2, 1.0 + u
    f = lambda y: (a * (y / beta) ** (2.0 / 3.0) - b) * kv(1.0 / 3.0, y)
    return quad(f, beta, np.inf, limit=300, epsabs=0, epsrel=1e-11)[0] / (1.0 + u) ** 3


def quantum_cdf(chi):
    # u grid (the sampler accepts any increasing abscissae): 10 points uniform in ln u from deep inside
    # the u^(1/3) power-law region (u << min(1, chi)) up to 0.3 chi, then 21 points u = 0.3 chi +
    # 21.7 chi s^1.7 (s uniform in (0, 1]) out to 22 chi, where the exponentially cut-off tail holds
    # < 1e-6 of the total.  Measured against quadrature this keeps interpolated quantiles within 0.6 %
    # of min(cdf, 1 - cdf) for cdf in [1e-3, 0.99] and within 3 % down to 1 - cdf = 1e-6; a grid
    # uniform in ln u is 30 % off in the tail.
    n_log = 10
    u_lo, u_break, u_hi = 1.0e-3 * min(1.0, chi), 0.3 * chi, 22.0 * chi
    s = np.linspace(0.0, 1.0, N - n_log + 1)[1:]
    ln_u = np.concatenate([np.linspace(np.log(u_lo), np.log(u_break), n_log),
                           np.log(u_break + (u_hi - u_break) * s ** 1.7)])
    g = lambda l: marginal(np.exp(l), chi) * np.exp(l)
    head = quad(g, -60.0, ln_u[0], limit=400, epsabs=0, epsrel=1e-10)[0]
    pieces = np.array([quad(g, a, b, limit=200, epsabs=0, epsrel=1e-11)[0] for a, b in zip(ln_u[:-1], ln_u[1:])])
    tail = quad(g, ln_u[-1], ln_u[-1] + 8.0, limit=200, epsabs=0, epsrel=1e-10)[0]
    total = head + pieces.sum() + tail
    below = head + np.concatenate([[0.0], np.cumsum(pieces)])              # int_0^u, summed upwards
    above = tail + np.concatenate([np.cumsum(pieces[::-1])[::-1], [0.0]])  # int_u^inf, summed downwards
    # ln cdf from whichever side is the smaller number, so that neither end loses digits
    ln_cdf = np.where(below < above, np.log(below / total), np.log1p(-above / total))
    coeff = np.exp(ln_cdf[0]) / np.exp(ln_u[0]) ** (1.0 / 3.0)
    return coeff, 1.0 / 3.0, np.column_stack([ln_u, ln_cdf])


def y_cdf(delta):
    if np.isinf(delta):
        pdf = lambda y: y ** (2.0 / 3.0) * kv(1.0 / 3.0, y)
        y0, power = 0.0, 4.0 / 3.0
    else:
        pdf = lambda y: (delta * y ** (2.0 / 3.0) - 1.0) * kv(1.0 / 3.0, y)
        y0, power = delta ** -1.5, 2.0
    # geometric spacing of (y - y0): dense where the cdf starts, out to where K_{1/3} has died
    first = 0.02 * min(1.0, max(y0, 0.02)) if y0 > 0 else 0.01
    off = first * (32.0 / first) ** (np.arange(N) / (N - 1.0))
    ys = y0 + off
    c = cumulative(pdf, ys, y0)
    total = c[-1] + quad(pdf, ys[-1], np.inf, limit=200, epsabs=0, epsrel=1e-10)[0]
    cdf = c / total
    coeff = cdf[0] / (ys[0] - y0) ** power
    return coeff, power, np.column_stack([ys, cdf])


def classical_spectrum():
    # 16 points uniform in ln x over [0.02, 0.8] (below 0.02 the sampler uses the x^(5/3) power law),
    # then 25 points x = 0.8 + 19.2 s^1.5 out to 20, for the same reason as in quantum_cdf
    s = np.linspace(0.0, 1.0, 26)[1:]
    x = np.concatenate([np.exp(np.linspace(np.log(0.02), np.log(0.8), 16)), 0.8 + 19.2 * s ** 1.5])
    f = lambda t: (3.0 / np.pi) * t * kv(1.0 / 3.0, t)
    below = cumulative(f, x, 0.0)
    pieces = np.array([quad(f, a, b, limit=200, epsabs=0, epsrel=1e-11)[0] for a, b in zip(x[:-1], x[1:])])
    tail = quad(f, x[-1], np.inf, limit=200, epsabs=0, epsrel=1e-10)[0]
    above = tail + np.concatenate([np.cumsum(pieces[::-1])[::-1], [0.0]])
    return np.column_stack([np.log(x), np.where(below < above, np.log(below), np.log1p(-above))])


def emit_cdf(f, name_or_none, coeff, power, tab):
    f.write("    {%.17e, %.17e, {\n" % (coeff, power))
    for r in tab:
        f.write("        {%.17e, %.17e},\n" % (r[0], r[1]))
    f.write("    }},\n")


def main():
    out = os.path.join(ROOT, "opal_amd", "csrc", "qed_tables.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/make_qed_tables.py from the quantum synchrotron rate; do not edit.\n")
        f.write("// struct QedCdf { double coeff, power; double table[31][2]; };\n")
        f.write("#define QED_LN_CHI_MIN %.17e\n#define QED_LN_CHI_STEP %.17e\n" % (np.log(0.01), np.log(10.0) / 10.0))
        f.write("#define QED_LN_DELTA_MIN %.17e\n#define QED_LN_DELTA_STEP %.17e\n" % (np.log(0.1), np.log(10.0) / 10.0))
        f.write("static QED_TABLE_QUAL QedCdf QED_QUANTUM_CDF[41] = {\n")
        for i in range(41):
            chi = 0.01 * 10.0 ** (i / 10.0)
            emit_cdf(f, None, *quantum_cdf(chi))
            print("quantum cdf", i, file=sys.stderr)
        f.write("};\nstatic QED_TABLE_QUAL QedCdf QED_Y_CDF[31] = {\n")
        for j in range(31):
            emit_cdf(f, None, *y_cdf(0.1 * 10.0 ** (j / 10.0)))
            print("y cdf", j, file=sys.stderr)
        f.write("};\nstatic QED_TABLE_QUAL QedCdf QED_Y_INFINITE_DELTA_CDF[1] = {\n")
        emit_cdf(f, None, *y_cdf(np.inf))
        f.write("};\nstatic QED_TABLE_QUAL double QED_CLASSICAL_SPECTRUM[41][2] = {\n")
        for r in classical_spectrum():
            f.write("    {%.17e, %.17e},\n" % (r[0], r[1]))
        f.write("};\n")
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
