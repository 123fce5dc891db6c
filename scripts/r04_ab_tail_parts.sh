#!/bin/bash
# Same-box A/B of the linked step's tail at a cut with thousands of leavers per step: a hot plasma (u_th = 1 mc) of 512 cells x
# (20000 + 20000) particles linked to itself through the mailboxes; the tail's sending and unpacking by single workgroups
# (OPAL_TAIL_PARTS_AT huge) against shared among 8 per species and side (OPAL_TAIL_PARTS_AT=1), alternating.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_tail_parts
mkdir -p $O
for rep in 1 2; do
 for at in 2000000000 1; do
  OPAL_BENCH_HOT=1.0 OPAL_TAIL_PARTS_AT=$at OPAL_BENCH_LINK=peer python3 $R/bench.py --self-exchange --cells 512 --npc 20000 --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-default-path 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('parts_at=$at', round(d['ms_per_step'],4), 'ms/step  push', round(r['avg_launch_ms'],4), ' tail', round(r['other_kernels_avg_ms']['field_solve'],4), 'ms  value %.4e' % d['value'], flush=True)" | tee -a $O/ab.txt
 done
done
