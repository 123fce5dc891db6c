#!/bin/bash
# Instruction-fetch counters of k_radiate (and of the photon push as a small-kernel control) over the colliding-beams run.
# usage (through gpurun, from the repo root): scripts/r05_radiate_icache.sh  ->  gpurun_out/r05_radiate_pmc_ifetch.json
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/radiate_if_raw
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-include-regex "k_radiate|k_push_photon" --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/a -o a -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/a.log || echo "pass a failed"
rocprofv3 --kernel-include-regex "k_radiate|k_push_photon" --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/b -o b -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/b.log || echo "pass b failed"
python3 - $O $R/gpurun_out/r05_radiate_pmc_ifetch.json <<'PY'
import collections, csv, glob, json, os, sys
raw, out = sys.argv[1:3]
res = {"note": "sums over every launch of scripts/bench_colliding_beams.py (3000 steps), per kernel; two --pmc passes (a, b)", "passes": {}}
for p in ("a", "b"):
    f = glob.glob(os.path.join(raw, p, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        res["passes"][p] = "no output: " + open(os.path.join(raw, p + ".log")).read()[-600:]
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = "k_radiate" if "k_radiate" in r["Kernel_Name"] else "k_push_photon"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res["passes"][p] = {k: dict(v) for k, v in acc.items()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["passes"], indent=1)[:4000])
PY
rm -rf $O
