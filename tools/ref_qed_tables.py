"""Reads the NUMBERS of the reference's photon-emission lookup tables, as data, at run time.

Build-container only: /root/reference does not travel to the GPU box, so nothing under `-m gpu`,
smoke() or bench.py imports this.  Two users:

  * tools/transcribe_qed_tables.py  writes opal_amd/csrc/qed_tables.inc from these numbers (constant data
    parity depends on, like the physical constants and the ln h(chi) rate table);
  * tests/golden/make_golden_qed_sample.py  evaluates an independent Python restatement of
    photon_emission::sample on them and commits inputs + outputs as golden vectors.

Sources (data only; no code is taken):
  /root/reference/src/qed/photon_emission_tables.rs:3-7     LN_CHI_MIN/STEP, LN_DELTA_MIN/STEP
  /root/reference/src/qed/photon_emission_tables.rs:15      QUANTUM_CDF[41]
  /root/reference/src/qed/photon_emission_tables.rs:1371    Y_INFINITE_DELTA_CDF
  /root/reference/src/qed/photon_emission_tables.rs:1406    Y_CDF[31]
  /root/reference/src/qed/photon_emission.rs:209            CLASSICAL_SPECTRUM_TABLE[41]
"""
import os
import re

REFERENCE = os.environ.get("OPAL_REFERENCE", "/root/reference")
TABLES_RS = os.path.join(REFERENCE, "src", "qed", "photon_emission_tables.rs")
EMISSION_RS = os.path.join(REFERENCE, "src", "qed", "photon_emission.rs")

_NUM = r"[-+]?\d+\.\d*(?:[eE][-+]?\d+)?"


def _ratio(text):
    """A Rust f64 constant expression of the form `a` or `a/b` (e.g. `1.0/3.0`), evaluated in f64."""
    parts = [float(p) for p in text.split("/")]
    assert 1 <= len(parts) <= 2, text
    return parts[0] if len(parts) == 1 else parts[0] / parts[1]


def _cdfs(block):
    """Every `CDF {coeff: c, power: p, table: [[x, f], ...]}` literal in `block`."""
    out = []
    for m in re.finditer(r"CDF\s*\{\s*coeff:\s*(%s)\s*,\s*power:\s*([^,]+),\s*table:\s*\[(.*?)\]\s*\}" % _NUM, block, re.S):
        rows = [(float(a), float(b)) for a, b in re.findall(r"\[\s*(%s)\s*,\s*(%s)\s*\]" % (_NUM, _NUM), m.group(3))]
        assert len(rows) == 31, len(rows)
        out.append({"coeff": float(m.group(1)), "power": _ratio(m.group(2).strip()), "table": rows})
    return out


def available():
    return os.path.exists(TABLES_RS) and os.path.exists(EMISSION_RS)


def load():
    """-> dict with the four scalar constants, 'quantum' (41 cdfs), 'y' (31), 'y_inf' (1), 'classical' (41 rows)."""
    src = open(TABLES_RS).read()
    consts = {}
    for name in ("LN_CHI_MIN", "LN_CHI_STEP", "LN_DELTA_MIN", "LN_DELTA_STEP"):
        m = re.search(r"pub const %s:\s*f64\s*=\s*(%s)\s*;" % (name, _NUM), src)
        consts[name] = float(m.group(1))
    iq = src.index("pub static QUANTUM_CDF")
    ii = src.index("pub static Y_INFINITE_DELTA_CDF")
    iy = src.index("pub static Y_CDF")
    assert iq < ii < iy
    quantum, y_inf, y = _cdfs(src[iq:ii]), _cdfs(src[ii:iy]), _cdfs(src[iy:])
    assert (len(quantum), len(y_inf), len(y)) == (41, 1, 31)
    em = open(EMISSION_RS).read()
    ic = em.index("= [", em.index("static CLASSICAL_SPECTRUM_TABLE"))
    body = em[ic:em.index("];", ic)]
    classical = [(float(a), float(b)) for a, b in re.findall(r"\[\s*(%s)\s*,\s*(%s)\s*\]" % (_NUM, _NUM), body)]
    assert len(classical) == 41
    m = re.search(r"(%s)\s*\*\s*rand1\.powf\(0\.6\)" % _NUM, em)
    consts["CLASSICAL_SMALL_X_COEFF"] = float(m.group(1))
    return dict(consts, quantum=quantum, y=y, y_inf=y_inf[0], classical=classical)
