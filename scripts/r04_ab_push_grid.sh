#!/bin/bash
# Same-box A/B: the paired push as a grid of one workgroup per chunk (OPAL_PUSH_GRID unset) against a fixed grid whose
# workgroups stride over the chunks (persistent workgroups), at one GPU's share of configs[2] at N = 8 and at configs[1].
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_push_grid
mkdir -p $O
COMMON="--elision on --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --warmup 50"
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],5), 'ms/step  push', round(d['roofline']['avg_launch_ms'],5), 'value %.4e' % d['value'], flush=True)" | tee -a $O/ab.txt; }
for rep in 1 2; do
 for g in 0 ${GRIDS:-1024 1536 2048 3072}; do
  OPAL_PUSH_GRID=$g python3 $R/bench.py $COMMON --self-exchange --cells 4096 --npc 1526 --steps 2000 2>>$O/err.log | line "share8 grid=$g"
  OPAL_PUSH_GRID=$g python3 $R/bench.py $COMMON --cells 4096 --npc 2441 --steps 1000 2>>$O/err.log | line "configs1 grid=$g"
 done
done
