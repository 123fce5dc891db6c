#!/bin/bash
# Kernel traces of the multi-GPU step rehearsed on one GPU (bench.py --self-exchange --elision on): one GPU's share of
# BASELINE configs[2] at N = 8 (4096 cells x 1526+1526), through the peer-mapped mailboxes and through RCCL.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-trace_linked}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for LINK in peer rccl; do
  OPAL_BENCH_LINK=$LINK rocprofv3 --kernel-trace --output-format csv -d $O/$LINK -o kt -- python3 $R/bench.py --self-exchange --elision on --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary --no-live-traffic --steps 400 --warmup 20 > $O/bench_$LINK.json 2> $O/$LINK.log
  python3 $R/scripts/step_timeline.py $O/$LINK 2 20 > $O/timeline_$LINK.txt
  rm -rf $O/$LINK
done
# and untraced, alternating (a trace costs a few per cent)
for LINK in peer rccl peer rccl; do
  OPAL_BENCH_LINK=$LINK python3 $R/bench.py --self-exchange --elision on --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary --steps 2000 --warmup 100 >> $O/untraced_$LINK.json 2>> $O/untraced.log
done
echo done
