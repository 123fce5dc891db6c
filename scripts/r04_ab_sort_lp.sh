#!/bin/bash
# Same-box A/B on the laser-plasma interaction (2e7 particles, contract form): the re-binning's scatter straight to the output
# (OPAL_SORT_STAGED=0) against staged through LDS (1), and the price the policy puts on a re-binning (ns per slot).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_sort_lp
mkdir -p $O
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],5), 'ms/step  push', round(r['avg_launch_ms'],5), 'rebins', r['rebins_in_timed_region'], 'x', round(r['other_kernels_avg_ms']['rebin'],4), 'field', round(r['other_kernels_avg_ms']['field_solve'],4), 'whole-step frac %.4f' % (d['value'] * 80 / 8e12), flush=True)" | tee -a $O/ab.txt; }
for rep in 1 2; do
 for cfg in ${CFGS:-"0 0.028" "1 0.028" "1 0.022" "1 0.018"}; do
  set -- $cfg
  OPAL_SORT_STAGED=$1 OPAL_SORT_NS_PER_SLOT=$2 python3 $R/bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path 2>>$O/err.log | line "staged=$1 ns_per_slot=$2"
 done
done
