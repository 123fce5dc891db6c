#!/bin/bash
# Same-box comparison of the linked step's tail kernel between library builds (opal_amd/lib/libvar_*.so and the working
# tree's): kernel trace of bench.py --self-exchange at one GPU's share of configs[2] at N = 8, peer mailboxes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_tail; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for round in 1 2; do for v in "$@"; do
  OPAL_AMD_LIB=$R/opal_amd/lib/$v.so OPAL_BENCH_LINK=peer rocprofv3 --kernel-trace --output-format csv -d $O/$v -o kt -- python3 $R/bench.py --self-exchange --elision on --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary --steps 600 --warmup 20 > $O/bench_$v.json 2> $O/$v.log
  echo "== $v"; python3 $R/scripts/step_timeline.py $O/$v 2 20 | head -5 | tail -4
  python3 - $O/$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_link_tail" in r["Kernel_Name"]]
d=d[-500:]; d.sort(); print("   k_link_tail over the last 500 launches: median %.2f us, p10 %.2f, p90 %.2f" % (d[len(d)//2]/1e3, d[len(d)//10]/1e3, d[9*len(d)//10]/1e3))
PY
  rm -rf $O/$v
done; done
