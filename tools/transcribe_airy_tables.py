#!/usr/bin/env python3
"""Writes opal_amd/csrc/airy_tables.inc: the reference's Airy-function constants as constant data.

airy_ai_for_positive (qed/special_functions/airy.rs:19-69) is a function of its constants as much as of its
code: the device and the oracle only return what the reference's arithmetic would give if SMALL_X_EXPANSION
(airy.rs:115-142), GAUSS_LAGUERRE_{40,16,4}_{NODES,WEIGHTS} (airy.rs:144-278) and the scale factor 0.262183997088323
(airy.rs:42,52,62) hold the reference's digits.  They are transcribed here the way constants.rs, the ln h(chi)
table and the photon-emission sampling tables are: numbers only, parsed from the reference's file at run time and
re-emitted in this repo's own layout with 17 significant digits (the reference prints 16; both parse to the same f64).

Run in the build container (needs /root/reference):  python tools/transcribe_airy_tables.py
tools/make_airy_tables.py stays as the mathematical cross-check (scipy: the same constants computed, agreeing to
4e-16 on the nodes and 2e-13 on the weights); tests/test_absorption_oracle.py compares both with the committed file.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/qed/special_functions/airy.rs"
NUM = r"[-+]?\d\.\d+e[-+]?\d+"


def load(path=REF):
    src = open(path).read()

    def arr(name):
        m = re.search(r"static %s: \[f64; (\d+)\] = \[(.*?)\];" % name, src, re.S)
        v = [float(s) for s in re.findall(NUM, m.group(2))]
        assert len(v) == int(m.group(1)), name
        return v
    i = src.index("static SMALL_X_EXPANSION")
    a = [float(s) for s in re.findall(NUM, src[src.index("a: [", i):src.index("],", i)])]
    j = src.index("n: [", i)
    n = [int(s) for s in re.findall(r"\d+", src[j + 4:src.index("]", j)])]
    assert len(a) == len(n) == 20
    used = sorted(set(int(s) for s in re.findall(r"evaluate_up_to\(x, (\d+)\)", src)))
    scale = sorted(set(re.findall(r"let a = (\d\.\d+) \* s\.powf", src)))
    assert len(used) == 1 and len(scale) == 1
    gl = {m: (arr("GAUSS_LAGUERRE_%d_NODES" % m), arr("GAUSS_LAGUERRE_%d_WEIGHTS" % m)) for m in (40, 16, 4)}
    return {"series_a": a, "series_n": n, "series_used": used[0], "scale": float(scale[0]), "gl": gl}


def render(t, f):
    f.write("// Constant data: the reference's Airy-function constants (numbers only), written by tools/transcribe_airy_tables.py "
            "from src/qed/special_functions/airy.rs:42,115-278.  Do not edit.\n")
    f.write("#define AIRY_SERIES_TERMS 20\n#define AIRY_SERIES_USED %d\n" % t["series_used"])
    f.write("#define AIRY_SCALE %.17e\n" % t["scale"])
    f.write("static AIRY_TABLE_QUAL double AIRY_SERIES_A[20] = {\n")
    for v in t["series_a"]:
        f.write("    %.17e,\n" % v)
    f.write("};\nstatic AIRY_TABLE_QUAL int AIRY_SERIES_N[20] = {%s};\n" % ", ".join(str(v) for v in t["series_n"]))
    for m in (40, 16, 4):
        x, w = t["gl"][m]
        f.write("static AIRY_TABLE_QUAL double AIRY_GL%d_X[%d] = {\n" % (m, m))
        for v in x:
            f.write("    %.17e,\n" % v)
        f.write("};\nstatic AIRY_TABLE_QUAL double AIRY_GL%d_W[%d] = {\n" % (m, m))
        for v in w:
            f.write("    %.17e,\n" % v)
        f.write("};\n")


def main(out=None):
    out = out or os.path.join(ROOT, "opal_amd", "csrc", "airy_tables.inc")
    with open(out, "w") as f:
        render(load(), f)
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:2])
