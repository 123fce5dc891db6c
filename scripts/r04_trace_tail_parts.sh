#!/bin/bash
# k_link_tail's duration at a cut with thousands of leavers per step (the hot self-linked plasma of r04_ab_tail_parts.sh), single
# workgroups against 8 per species and side: rocprofv3 --kernel-trace --stats of both.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_trace_tail_parts
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for at in 2000000000 1; do
  OPAL_BENCH_HOT=1.0 OPAL_TAIL_PARTS_AT=$at OPAL_BENCH_LINK=peer rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$at -o kt -- python3 $R/bench.py --self-exchange --cells 512 --npc 20000 --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-default-path --no-live-traffic > $O/bench_$at.json 2> $O/log_$at.txt
  echo "== OPAL_TAIL_PARTS_AT=$at" >> $O/summary.txt
  python3 - $O/kt_$at >> $O/summary.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print("  %-70s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -rf $O/kt_$at
done
cat $O/summary.txt
