#!/usr/bin/env python3
"""Generates tests/golden/qed_emission_vectors.json: fixed inputs / outputs of the photon-emission path.

Like pic_step_vectors.json these are restatement-generated (the Rust reference cannot run here): they come
from the CPU oracle, whose pwmci / vec3 / rate functions are pinned by the reference's known-answer tests
and whose sampler is pinned bit for bit by qed_sample_reference_tables.json (an independent restatement on the
reference's own tables, make_golden_qed_sample.py).  They freeze the oracle and the transcribed sampling
tables (opal_amd/csrc/qed_tables.inc), and give the GPU tests inputs and outputs that do not depend on the
oracle at run time.

Contents
  philox   : Random123's published known answers for philox4x32-10
  randoms  : the four uniforms of an emission for a few (seed, id, step)
  rate     : photon_emission::rate (photon_emission.rs:59-79) across its three branches
  sample   : photon_emission::sample (photon_emission.rs:129-203) for chi in [0.0101, 99]
  classical: classical_sample (photon_emission.rs:264-292)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402


def main():
    o.build()
    out = {"_generator": "tests/golden/make_golden_qed.py (CPU oracle; the Rust reference cannot run here)"}
    out["philox"] = [
        {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]},
        {"ctr": [0xFFFFFFFF] * 4, "key": [0xFFFFFFFF] * 2, "out": [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]},
        {"ctr": [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], "key": [0xA4093822, 0x299F31D0],
         "out": [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]},
    ]
    out["randoms"] = [{"seed": s, "id": i, "step": t, "out": [float(v) for v in o.emission_randoms(s, i, t)]}
                      for s, i, t in ((0, 0, 0), (42, 7, 3), (0x6F70616C5F616D64, (3 << 40) + 12345, 1234567),
                                      (2 ** 64 - 1, 2 ** 63 + 5, 2 ** 32 - 1))]
    out["rate"] = [{"chi": c, "gamma": g, "out": o.rate(c, g)}
                   for c, g in ((1e-5, 10.0), (0.0099, 50.0), (0.01, 50.0), (0.026, 1000.0), (0.5, 300.0), (3.5, 1000.0),
                                (9.98, 1000.0), (12.4, 1000.0), (99.99, 2000.0), (100.0, 2000.0), (403.0, 1000.0))]
    rng = np.random.default_rng(77)
    smp = []
    for k in range(60):
        chi = float(np.exp(rng.uniform(np.log(0.0101), np.log(99.0))))
        gamma = float(np.exp(rng.uniform(np.log(20.0), np.log(2.0e4))))
        r = [float(v) for v in rng.random(3)]
        if k % 10 == 0:
            r[0] = float(10.0 ** rng.uniform(-10, -4))       # power-law head of the energy cdf
        if k % 10 == 1:
            r[0] = float(1.0 - 10.0 ** rng.uniform(-9, -4))  # tail beyond / near the last tabulated point
        if k % 10 == 2:
            r[1] = float(10.0 ** rng.uniform(-10, -4))
        if k % 10 == 3:
            r[1] = float(1.0 - 10.0 ** rng.uniform(-9, -4))
        smp.append({"chi": chi, "gamma": gamma, "r": r, "out": [float(v) for v in o.qed_sample(chi, gamma, *r)]})
    out["sample"] = smp
    cl = []
    for k in range(20):
        chi = float(np.exp(rng.uniform(np.log(1e-5), np.log(0.0099))))
        gamma = float(np.exp(rng.uniform(np.log(20.0), np.log(2.0e4))))
        r = [float(v) for v in rng.random(3)]
        if k % 5 == 0:
            r[0] = float(10.0 ** rng.uniform(-10, -4))
        cl.append({"chi": chi, "gamma": gamma, "r": r, "out": [float(v) for v in o.qed_classical_sample(chi, gamma, *r)]})
    out["classical"] = cl
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qed_emission_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
