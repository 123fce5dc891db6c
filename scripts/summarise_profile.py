#!/usr/bin/env python3
"""Turns one run of scripts/profile_bench.sh (gpurun_out/<dir>) into the tracked summaries under profiles/:
  <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the bench command (per-kernel calls / total / average)
  <tag>_bench_under_rocprof.json   the bench line of that same profiled run
  <tag>_pmc_hbm.json               FETCH_SIZE / WRITE_SIZE per launch of every push kernel instantiation that ran
                                   (separate passes), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts half
                                   of a coalesced streaming read)
  <tag>_pmc_sq.json                SQ / GRBM counters per launch of the same kernels and what follows from them
  traffic.json                     HBM bytes per particle-update, keyed "<workload_key>|<kernel instantiation>", with the
                                   hash of the kernel sources they were taken with; bench.py reports an entry as
                                   roofline.traffic only for the kernel it timed, on that workload, with those sources
usage: summarise_profile.py gpurun_out/<dir> <tag> [--last N]
  --last N: only the last N launches of each kernel count (the timed steps of a run whose warm-up is a different phase
            of the physics: the laser-plasma interaction behind 21 000 quiet steps); kernel times then come from the
            kernel trace instead of the run's --stats table
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
LAST = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (OPAL_PROFILES_DIR: summarise on the GPU box into a scratch directory -- the raw counter files of a 23 000-step run
# are too big to bring back -- and copy the summaries into profiles/ afterwards; traffic.json entries are merged)
P = os.environ.get("OPAL_PROFILES_DIR") or os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
PUSH = re.compile(r"k_push(?:_pair)?<[^>]*>")

bench = json.load(open(os.path.join(src, "bench_under_kt.json")))
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
if LAST:
    open(os.path.join(P, f"{tag}_kernel_stats.csv"), "a").write(
        f"# (whole run, warm-up included; the summaries beside this file use the last {LAST} launches of each kernel)\n")
json.dump(bench, open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
n = bench["config"]["particles_total"]
wkey = bench["config"]["workload_key"]
sha = bench["config"]["kernel_src_sha"]

stats = {}
for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))):
    m = PUSH.search(r["Name"])
    if m:
        stats[m.group(0)] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
if LAST:
    durs = collections.defaultdict(list)
    rows = sorted(csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        m = PUSH.search(r["Kernel_Name"])
        if m:
            durs[m.group(0)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in durs.items():
        v = v[-LAST:]
        stats[k] = {"calls": len(v), "avg_ns": sum(v) / len(v), "of_the_last": LAST}
print("push kernels in the run:", json.dumps(stats, indent=1))


def per_launch(path):
    """{kernel instantiation: {counter: {launches, mean}}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(path)))
    if LAST:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        m = PUSH.search(r["Kernel_Name"])
        if m:
            acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if LAST:
        acc = {k: {c: v[-LAST:] for c, v in d.items()} for k, d in acc.items()}
    return {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}


f = per_launch(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"))
w = per_launch(os.path.join(src, "pmc_write", "w_counter_collection.csv"))
tpath = os.path.join(P, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic = {k: v for k, v in traffic.items() if "|" in k}  # (earlier rounds' layout: dropped)
hbm = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench command recorded in "
               f"{tag}_bench_under_rocprof.json (arguments of the counter passes: "
               + open(os.path.join(src, "args_pmc.txt")).read().strip() + "), per launch of each push kernel instantiation "
               "(one launch = both species). Units KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts "
               "half of a coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for "
               "streaming stores.",
       "workload_key": wkey, "kernel_src_sha": sha, "particle_updates_per_launch": n, "kernels": {}}
for k in sorted(set(f) & set(w)):
    rd = 2.0 * f[k]["FETCH_SIZE"]["mean"] * 1024.0
    wr = w[k]["WRITE_SIZE"]["mean"] * 1024.0
    avg_ns = stats.get(k, {}).get("avg_ns")
    hbm["kernels"][k] = {
        "FETCH_SIZE_KiB": f[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[k]["WRITE_SIZE"],
        "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "read_bytes_per_particle_update": rd / n, "write_bytes_per_particle_update": wr / n,
        "hbm_bytes_per_particle_update": (rd + wr) / n,
        "kernel_avg_ms_rocprof": avg_ns / 1e6 if avg_ns else None,
        "kernel_calls_rocprof": stats.get(k, {}).get("calls"),
        "hbm_GBps_actual": (rd + wr) / (avg_ns * 1e-9) / 1e9 if avg_ns else None,
        "frac_of_8TBps": (rd + wr) / (avg_ns * 1e-9) / 8e12 if avg_ns else None,
    }
    traffic[f"{wkey}|{k}"] = {"hbm_bytes_per_particle_update": (rd + wr) / n, "source": f"profiles/{tag}_pmc_hbm.json",
                              "src_sha": sha}
    print(k, "B/update", (rd + wr) / n, "GB/s", hbm["kernels"][k]["hbm_GBps_actual"])
json.dump(hbm, open(os.path.join(P, f"{tag}_pmc_hbm.json"), "w"), indent=1)
json.dump(traffic, open(tpath, "w"), indent=1)

sq_path = os.path.join(src, "pmc_sq", "sq_counter_collection.csv")
if os.path.exists(sq_path):
    s1 = per_launch(sq_path)
    s2 = per_launch(os.path.join(src, "pmc_sq2", "sq2_counter_collection.csv"))
    sq = {"note": "per launch of each push kernel instantiation; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles "
                  "summed over waves or SIMDs (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs",
          "workload_key": wkey, "kernel_src_sha": sha, "kernels": {}}
    for k in sorted(set(s1) & set(s2)):
        c = {kk: v["mean"] for kk, v in {**s1[k], **s2[k]}.items()}
        xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        avg_ns = stats.get(k, {}).get("avg_ns")
        sq["kernels"][k] = {
            "counters": c,
            "valu_instructions_per_wave_particle": c["SQ_INSTS_VALU"] / (n / 64.0),
            "salu_instructions_per_wave_particle": c["SQ_INSTS_SALU"] / (n / 64.0),
            "lds_instructions_per_wave_particle": c["SQ_INSTS_LDS"] / (n / 64.0),
            "valu_busy_fraction_per_simd": 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / xcd_cycles,
            "wave_cycles_split": {kk: c[kk] / c["SQ_WAVE_CYCLES"] for kk in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
            "lds_issue_stall_fraction_of_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_fraction_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
            "mean_waves_per_simd_resident": c["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / xcd_cycles,
            "effective_clock_GHz": xcd_cycles / (avg_ns * 1e-9) / 1e9 if avg_ns else None,
        }
        print(k, json.dumps({kk: v for kk, v in sq["kernels"][k].items() if kk != "counters"}, indent=1))
    json.dump(sq, open(os.path.join(P, f"{tag}_pmc_sq.json"), "w"), indent=1)
