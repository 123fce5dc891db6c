#!/bin/bash
# SQ counters of k_radiate over the whole colliding-beams run (scripts/bench_colliding_beams.py, 1e7 electrons, 3000 steps): how many of
# a wave's 64 lanes an issued VALU instruction has active (the divergence of Brent's loop, tests/test_qed_oracle.py::
# test_brent_iterations_have_a_long_tail) and how busy the VALUs are.  Counter passes only (no trace domain beside --pmc), python3 itself
# behind `--`.  usage (through gpurun, from the repo root): scripts/r05_radiate_counters.sh   ->  gpurun_out/r05_radiate_pmc_sq.json
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/radiate_raw
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-include-regex "k_radiate" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o a -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/a.log || echo "pass a failed"
rocprofv3 --kernel-include-regex "k_radiate" --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/b -o b -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/b.log || echo "pass b failed"
python3 - $O $R/gpurun_out/r05_radiate_pmc_sq.json <<'PY'
import collections, csv, glob, json, os, sys
raw, out = sys.argv[1:3]
res = {"note": "k_radiate, sums over every launch of scripts/bench_colliding_beams.py (3000 steps); SQ_*_CYCLES / SQ_ACTIVE_INST_* count "
               "quad-cycles summed over waves or SIMDs (MI355X_MICROARCH.md); two --pmc passes (a, b)", "passes": {}}
for p in ("a", "b"):
    f = glob.glob(os.path.join(raw, p, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        res["passes"][p] = "no output: " + open(os.path.join(raw, p + ".log")).read()[-400:]
        continue
    acc, launches = collections.defaultdict(float), set()
    for r in csv.DictReader(open(f[0])):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); launches.add(r.get("Dispatch_Id"))
    res["passes"][p] = dict(acc, launches=len(launches))
a, b = res["passes"].get("a"), res["passes"].get("b")
if isinstance(a, dict) and a.get("SQ_WAVES"):
    res["valu_instructions_per_wave"] = a["SQ_INSTS_VALU"] / a["SQ_WAVES"]
    res["salu_instructions_per_wave"] = a["SQ_INSTS_SALU"] / a["SQ_WAVES"]
    res["valu_busy_fraction_of_busy_cycles_per_simd"] = 4.0 * a["SQ_ACTIVE_INST_VALU"] / (a["SQ_BUSY_CYCLES"] / 8 * 1024) if a.get("SQ_BUSY_CYCLES") else None
    res["wave_cycles_per_valu_instruction"] = 4.0 * a["SQ_WAVE_CYCLES"] / a["SQ_INSTS_VALU"]
if isinstance(b, dict) and b.get("SQ_THREAD_CYCLES_VALU") and b.get("SQ_ACTIVE_INST_VALU"):
    res["active_lanes_per_valu_instruction_of_64"] = b["SQ_THREAD_CYCLES_VALU"] / b["SQ_ACTIVE_INST_VALU"]
    res["active_lanes_alt_per_insts_valu"] = b["SQ_THREAD_CYCLES_VALU"] / b["SQ_INSTS_VALU"] if b.get("SQ_INSTS_VALU") else None
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("note",)}, indent=1)[:3000])
PY
rm -rf $O
