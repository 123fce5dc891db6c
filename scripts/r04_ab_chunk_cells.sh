#!/bin/bash
# Same-box sweep of the most cells a chunk of the table may span (OPAL_CHUNK_CELLS; 24 since round 2) on the laser-plasma
# interaction (2e7 particles, contract form), and a check on the two-stream contract run.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_chunk_cells
mkdir -p $O
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],5), 'ms/step  push', round(r['avg_launch_ms'],5), 'rebins', r.get('rebins_in_timed_region'), 'x', round(r.get('other_kernels_avg_ms',{}).get('rebin',0),4), 'whole-step frac %.4f' % (d['value'] * 80 / 8e12), flush=True)" | tee -a $O/ab.txt; }
for rep in 1 2; do
 for cc in ${CCS:-24 16 8 4}; do
  OPAL_CHUNK_CELLS=$cc python3 $R/bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path 2>>$O/err.log | line "lp chunk_cells=$cc"
 done
done
for cc in 24 8; do
 OPAL_CHUNK_CELLS=$cc python3 $R/bench.py --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --no-default-path --steps 200 2>>$O/err.log | line "two-stream chunk_cells=$cc"
done
