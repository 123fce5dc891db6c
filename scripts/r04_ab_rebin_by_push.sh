#!/bin/bash
# Same-box A/B on the laser-plasma interaction (2e7 particles, contract form): re-binnings as passes of their own
# (OPAL_REBIN_BY_PUSH=0) against re-binnings made by the next paired push (1), and the price the policy puts on one.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_rebin_by_push
mkdir -p $O
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],5), 'ms/step  push', round(r['avg_launch_ms'],5), 'rebins', r['rebins_in_timed_region'], 'x', round(r['other_kernels_avg_ms']['rebin'],4), 'field', round(r['other_kernels_avg_ms']['field_solve'],4), 'whole-step frac %.4f' % (d['value'] * 80 / 8e12), flush=True)" | tee -a $O/ab.txt; }
for rep in 1 2; do
 for cfg in ${CFGS:-"0 0.028" "1 0.028" "1 0.012" "1 0.006"}; do
  set -- $cfg
  OPAL_REBIN_BY_PUSH=$1 OPAL_SORT_NS_PER_SLOT=$2 python3 $R/bench.py --workload laser_plasma --npc ${NPC:-2000} --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path 2>>$O/err.log | line "by_push=$1 ns_per_slot=$2"
 done
done
