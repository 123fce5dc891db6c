#!/usr/bin/env python3
"""Physics cross-check of the photon-emission sampling tables (NOT their source any more).

opal_amd/csrc/qed_tables.inc holds the reference's numbers (tools/transcribe_qed_tables.py).  This module
recomputes, by direct quadrature, what those tables tabulate, at the abscissae the tables use, so that a
transcription slip (a shifted row, a table attached to the wrong chi) shows up as a physics error:

with u = omega / (gamma m - omega), beta = 2u / (3 chi) and z = [2 gamma^2 (1 - beta_e cos(theta))]^(3/2) >= 1
the double-differential emission rate is

    d^2 N / (du dz)  ~  beta / (1+u)^3 * [ (1 + (1+u)^2) z^(2/3) - (1+u) ] * K_{1/3}(beta z).

  QUANTUM_CDF[i]   ln cdf(u; chi_i) against ln u, the marginal of the above over z in [1, inf),
                   chi_i = 0.01 * 10^(i/10), i = 0..40; below the first row cdf = coeff * u^(1/3)
  Y_CDF[j]         cdf of y = beta z for fixed delta_j = (1 + (1+u)^2) beta^(-2/3) / (1+u)
                   = 0.1 * 10^(j/10), j = 0..30:  pdf ~ (delta y^(2/3) - 1) K_{1/3}(y) on
                   [delta^(-3/2), inf); below the first row cdf = coeff * (y - delta^(-3/2))^2
  Y_INFINITE_DELTA the delta -> inf limit, pdf ~ y^(2/3) K_{1/3}(y) on [0, inf), cdf = coeff * y^(4/3)
  CLASSICAL_SPECTRUM   ln cdf(x) against ln x (0.02 <= x <= 20) for the classical conditional spectrum,
                   cdf(x) = (3/pi) int_0^x t K_{1/3}(t) dt  (x = 2 u z / (3 chi))

tests/test_qed_oracle.py::test_reference_tables_agree_with_quadrature uses these functions on the committed
numbers.  `python tools/make_qed_tables.py` prints the worst deviation of every table (a few minutes).
Measured: the reference's ln cdf(u) values are within 1e-7 of quadrature (they carry 9 digits), with a noise
floor of ~5e-12 in the far tails (1 - cdf < 1e-11, where the reference's rows are not even monotone); the y
tables within 1e-8 once the truncation at their last row is accounted for (see y_cdf); the classical table
within 2e-8; the power-law head coefficients are 0.1-0.2 % above the u -> 0 limit evaluated at the first row.
"""
import os
import re
import sys

import numpy as np
from scipy.integrate import quad
from scipy.special import kv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def marginal(u, chi):
    """dN/du up to a constant: the z-integral of the double-differential rate."""
    beta = 2.0 * u / (3.0 * chi)
    a, b = 1.0 + (1.0 + u) ** 2, 1.0 + u
    f = lambda y: (a * (y / beta) ** (2.0 / 3.0) - b) * kv(1.0 / 3.0, y)
    return quad(f, beta, np.inf, limit=300, epsabs=0, epsrel=1e-11)[0] / (1.0 + u) ** 3


def _two_sided(g, pts, lo, hi):
    """ln of the normalised integral of g from lo to each of the increasing pts, taken from whichever side is
    the smaller number so that neither end loses digits."""
    head = quad(g, lo, pts[0], limit=400, epsabs=0, epsrel=1e-10)[0]
    pieces = np.array([quad(g, a, b, limit=200, epsabs=0, epsrel=1e-11)[0] for a, b in zip(pts[:-1], pts[1:])])
    tail = quad(g, pts[-1], hi, limit=200, epsabs=0, epsrel=1e-10)[0]
    total = head + pieces.sum() + tail
    below = head + np.concatenate([[0.0], np.cumsum(pieces)])
    above = tail + np.concatenate([np.cumsum(pieces[::-1])[::-1], [0.0]])
    return np.where(below < above, np.log(below / total), np.log1p(-above / total))


def quantum_ln_cdf(chi, ln_u):
    """ln cdf(u; chi) at the given increasing ln u."""
    g = lambda l: marginal(np.exp(l), chi) * np.exp(l)
    return _two_sided(g, np.asarray(ln_u), -60.0, ln_u[-1] + 12.0)


def y_cdf(delta, ys):
    """cdf of y at the given increasing ys (delta = inf: the limiting table).  The reference's finite-delta
    tables are those of the distribution TRUNCATED at their last row (its cdf there is exactly 1; the mass
    beyond, 1.5e-3..4e-3, is dropped), the infinite-delta one is not: reproduced here."""
    if np.isinf(delta):
        pdf, y0 = (lambda y: y ** (2.0 / 3.0) * kv(1.0 / 3.0, y)), 0.0
        return np.exp(_two_sided(pdf, np.asarray(ys), y0, np.inf))
    pdf, y0 = (lambda y: (delta * y ** (2.0 / 3.0) - 1.0) * kv(1.0 / 3.0, y)), delta ** -1.5
    c = np.exp(_two_sided(pdf, np.asarray(ys), y0, np.inf))
    return c / c[-1]


def classical_ln_cdf(ln_x):
    f = lambda t: (3.0 / np.pi) * t * kv(1.0 / 3.0, t)
    return _two_sided(f, np.exp(np.asarray(ln_x)), 0.0, np.inf)


def committed_tables(path=None):
    """Parses opal_amd/csrc/qed_tables.inc (this repo's file, not the reference) into python lists."""
    text = open(path or os.path.join(ROOT, "opal_amd", "csrc", "qed_tables.inc")).read()
    num = r"[-+]?\d\.\d+e[-+]\d+"

    def cdfs(block):
        out = []
        for m in re.finditer(r"\{(%s), (%s), \{(.*?)\}\}," % (num, num), block, re.S):
            rows = np.array(re.findall(r"\{(%s), (%s)\}" % (num, num), m.group(3)), dtype=float)
            out.append((float(m.group(1)), float(m.group(2)), rows))
        return out
    iq, iy = text.index("QED_QUANTUM_CDF[41]"), text.index("QED_Y_CDF[31]")
    ii, ic = text.index("QED_Y_INFINITE_DELTA_CDF[1]"), text.index("QED_CLASSICAL_SPECTRUM[41][2]")
    classical = np.array(re.findall(r"\{(%s), (%s)\}" % (num, num), text[ic:]), dtype=float)
    return {"quantum": cdfs(text[iq:iy]), "y": cdfs(text[iy:ii]), "y_inf": cdfs(text[ii:ic])[0], "classical": classical}


def main():
    import warnings
    warnings.filterwarnings("ignore")
    t = committed_tables()
    for i, (coeff, power, rows) in enumerate(t["quantum"]):
        d = np.abs(quantum_ln_cdf(0.01 * 10.0 ** (i / 10.0), rows[:, 0]) - rows[:, 1])
        print("QUANTUM_CDF[%2d]  max |d ln cdf| = %.2e" % (i, d.max()))
    for j, (coeff, power, rows) in enumerate(t["y"]):
        d = np.abs(y_cdf(0.1 * 10.0 ** (j / 10.0), rows[:, 0]) - rows[:, 1])
        print("Y_CDF[%2d]        max |d cdf|    = %.2e" % (j, d.max()))
    coeff, power, rows = t["y_inf"]
    print("Y_INFINITE_DELTA max |d cdf|    = %.2e" % np.abs(y_cdf(np.inf, rows[:, 0]) - rows[:, 1]).max())
    print("CLASSICAL        max |d ln cdf| = %.2e" % np.abs(classical_ln_cdf(t["classical"][:, 0]) - t["classical"][:, 1]).max())


if __name__ == "__main__":
    sys.exit(main())
