#!/usr/bin/env python3
"""Writes opal_amd/csrc/qed_tables.inc: the reference's photon-emission sampling tables as constant data.

photon_emission::sample (photon_emission.rs:129-203) is a function of its tables as much as of its code:
identical (chi, gamma, r1, r2, r3) only give the reference's photon if QUANTUM_CDF, Y_CDF,
Y_INFINITE_DELTA_CDF and CLASSICAL_SPECTRUM_TABLE hold the reference's numbers.  They are transcribed
here the way constants.rs and the ln h(chi) rate table are: numbers only, parsed at run time by
tools/ref_qed_tables.py (which cites the lines), re-emitted in this repo's own struct layout with 17
significant digits (the reference prints 9; both parse to the same f64).

Run in the build container (needs /root/reference):  python tools/transcribe_qed_tables.py
tests/test_qed_oracle.py::test_committed_tables_are_the_reference_numbers re-parses the reference (when
present) and compares; tools/make_qed_tables.py remains as the physics cross-check (quadrature).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_qed_tables as ref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cdf(f, c):
    f.write("    {%.17e, %.17e, {\n" % (c["coeff"], c["power"]))
    for x, v in c["table"]:
        f.write("        {%.17e, %.17e},\n" % (x, v))
    f.write("    }},\n")


def render(t, f):
    f.write("// Constant data: the reference's photon-emission lookup tables (numbers only), written by\n"
            "// tools/transcribe_qed_tables.py from src/qed/photon_emission_tables.rs:3-7,15,1371,1406 and\n"
            "// src/qed/photon_emission.rs:209,275.  Do not edit.\n")
    f.write("// struct QedCdf { double coeff, power; double table[31][2]; };\n")
    f.write("#define QED_LN_CHI_MIN %.17e\n#define QED_LN_CHI_STEP %.17e\n" % (t["LN_CHI_MIN"], t["LN_CHI_STEP"]))
    f.write("#define QED_LN_DELTA_MIN %.17e\n#define QED_LN_DELTA_STEP %.17e\n" % (t["LN_DELTA_MIN"], t["LN_DELTA_STEP"]))
    f.write("#define QED_CLASSICAL_SMALL_X_COEFF %.17e\n" % t["CLASSICAL_SMALL_X_COEFF"])
    f.write("static QED_TABLE_QUAL QedCdf QED_QUANTUM_CDF[41] = {\n")
    for c in t["quantum"]:
        _cdf(f, c)
    f.write("};\nstatic QED_TABLE_QUAL QedCdf QED_Y_CDF[31] = {\n")
    for c in t["y"]:
        _cdf(f, c)
    f.write("};\nstatic QED_TABLE_QUAL QedCdf QED_Y_INFINITE_DELTA_CDF[1] = {\n")
    _cdf(f, t["y_inf"])
    f.write("};\nstatic QED_TABLE_QUAL double QED_CLASSICAL_SPECTRUM[41][2] = {\n")
    for x, v in t["classical"]:
        f.write("    {%.17e, %.17e},\n" % (x, v))
    f.write("};\n")


def main(out=None):
    out = out or os.path.join(ROOT, "opal_amd", "csrc", "qed_tables.inc")
    with open(out, "w") as f:
        render(ref.load(), f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(*sys.argv[1:2])
