#!/usr/bin/env python3
"""Turns one run of scripts/r02_profile.sh (gpurun_out/<dir>) into the tracked summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py` (per-kernel calls / total / average)
  <tag>_bench_under_rocprof.json   the bench line of that same profiled run
  <tag>_pmc_hbm.json       FETCH_SIZE / WRITE_SIZE per launch of the fused kernel (separate passes), corrected as
                           MI355X_MICROARCH.md prescribes (FETCH_SIZE counts half of a coalesced streaming read)
  <tag>_pmc_sq.json        SQ / GRBM counters per launch of the fused kernel and what follows from them
  traffic.json             HBM bytes per particle-update, read by bench.py for roofline.traffic
usage: summarise_profile.py gpurun_out/r02_prof_b r02 [traffic-key]     (traffic-key: two_stream (default),
       two_stream_no_elision, laser_plasma ... -- the entry of traffic.json this run describes)
"""
import collections
import re
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tkey = sys.argv[3] if len(sys.argv) > 3 else "two_stream"
# k_push_pair<FUSED, RHO, ES>: ES = the electrostatic launch; RHO = false on all but the last step of a
# sim.step() call.  The variant with the most launches in the run is the one summarised.
calls = collections.Counter()
for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))):
    if "k_push_pair<1, " in r["Name"]:
        calls[re.search(r"k_push_pair<[^>]*>", r["Name"]).group(0)] = int(r["Calls"])
KERNEL = calls.most_common(1)[0][0]
print("summarising", KERNEL, dict(calls))


def per_launch(path):
    acc = collections.defaultdict(list)
    name = None
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    return name, {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}


shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
bench = json.load(open(os.path.join(src, "bench_under_kt.json")))
json.dump(bench, open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
n = bench["config"]["particles_total"]
avg_ns = None
for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))):
    if KERNEL in r["Name"]:
        avg_ns = float(r["AverageNs"])

name, f = per_launch(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"))
_, w = per_launch(os.path.join(src, "pmc_write", "w_counter_collection.csv"))
rd = 2.0 * f["FETCH_SIZE"]["mean"] * 1024.0
wr = w["WRITE_SIZE"]["mean"] * 1024.0
hbm = {
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 5 --warmup 2 "
            "--no-cpu-baseline --no-secondary` plus the arguments recorded in the bench line beside this file, per "
            "launch of the fused kernel (one launch = both species). Units KiB. gfx950 correction "
            "(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half of a coalesced streaming read, so read bytes = "
            "2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for streaming stores. Bytes below the algorithmic 44 + 36 "
            "per update are what the kernel does not move: a uniform weight, uy / uz of an electrostatic run, "
            "write-back of attributes no lane of a wave changed (DESIGN.md 4.1).",
    "kernel": name,
    "FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"],
    "particle_updates_per_launch": n,
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": 80.0 * n,
    "read_bytes_per_particle_update": rd / n, "write_bytes_per_particle_update": wr / n,
    "hbm_bytes_per_particle_update": (rd + wr) / n,
    "kernel_avg_ms_rocprof": avg_ns / 1e6 if avg_ns else None,
    "hbm_GBps_actual": (rd + wr) / (avg_ns * 1e-9) / 1e9 if avg_ns else None,
}
json.dump(hbm, open(os.path.join(P, f"{tag}_pmc_hbm.json"), "w"), indent=1)
tpath = os.path.join(P, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
if "hbm_bytes_per_particle_update" in traffic:  # the single-entry layout of earlier rounds
    traffic = {}
traffic[tkey] = {"hbm_bytes_per_particle_update": (rd + wr) / n, "source": f"profiles/{tag}_pmc_hbm.json"}
json.dump(traffic, open(tpath, "w"), indent=1)

_, s1 = per_launch(os.path.join(src, "pmc_sq", "sq_counter_collection.csv"))
_, s2 = per_launch(os.path.join(src, "pmc_sq2", "sq2_counter_collection.csv"))
c = {k: v["mean"] for k, v in {**s1, **s2}.items()}
xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
sq = {
    "note": "per launch of the fused kernel; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
            "waves or SIMDs (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs",
    "counters": c,
    "valu_instructions_per_wave_particle": c["SQ_INSTS_VALU"] / (n / 64.0),
    "salu_instructions_per_wave_particle": c["SQ_INSTS_SALU"] / (n / 64.0),
    "lds_instructions_per_wave_particle": c["SQ_INSTS_LDS"] / (n / 64.0),
    "valu_busy_fraction_per_simd": 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / xcd_cycles,
    "wave_cycles_split": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
    "lds_issue_stall_fraction_of_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"],
    "lds_bank_conflict_fraction_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
    "mean_waves_per_simd_resident": c["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / xcd_cycles,
    "effective_clock_GHz": xcd_cycles / (avg_ns * 1e-9) / 1e9 if avg_ns else None,
}
json.dump(sq, open(os.path.join(P, f"{tag}_pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in sq.items() if k not in ("note", "counters")}, indent=1))
print("hbm B/update", hbm["hbm_bytes_per_particle_update"], "actual GB/s", hbm["hbm_GBps_actual"])
