#!/bin/bash
# SQ counters of k_radiate over the colliding-beams run (what the kernel waits for).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/radiate_pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-include-regex "k_radiate" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/a -o a -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/a.log
rocprofv3 --kernel-include-regex "k_radiate" --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/b -o b -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/b.log
python3 - $O <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    f = glob.glob(sys.argv[1] + f"/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print("no file", sub); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        late = v[len(v)*2//3: len(v)*5//6]   # launches inside the collision
        print("%-22s launches %5d  mean(all) %14.1f  mean(collision) %14.1f" % (k, len(v), sum(v)/len(v), sum(late)/max(len(late),1)))
PY
rm -rf $O/a $O/b
