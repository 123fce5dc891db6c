#!/bin/bash
# Same-box A/B of the electrostatic launch's chunk length: the round-3 rule (OPAL_PPT_ROUNDS=0) against whole rounds of
# the chip (1), alternating, at one GPU's share of configs[2] at N = 8 (linked through the mailboxes), configs[1] and configs[2].
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_ppt
mkdir -p $O
COMMON="--elision on --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --warmup 50"
for rep in 1 2; do
 for m in 0 1; do
  OPAL_PPT_ROUNDS=$m python3 $R/bench.py $COMMON --self-exchange --cells 4096 --npc 1526 --steps 2000 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share8 rounds=$m', round(d['ms_per_step'],5), 'ms/step  push', round(d['roofline']['avg_launch_ms'],5), 'value %.4e' % d['value'], flush=True)" | tee -a $O/ab.txt
  OPAL_PPT_ROUNDS=$m python3 $R/bench.py $COMMON --cells 4096 --npc 2441 --steps 1000 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs1 rounds=$m', round(d['ms_per_step'],5), 'ms/step  push', round(d['roofline']['avg_launch_ms'],5), 'value %.4e' % d['value'], flush=True)" | tee -a $O/ab.txt
  OPAL_PPT_ROUNDS=$m python3 $R/bench.py $COMMON --steps 300 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs2 rounds=$m', round(d['ms_per_step'],5), 'ms/step  push', round(d['roofline']['avg_launch_ms'],5), 'value %.4e' % d['value'], flush=True)" | tee -a $O/ab.txt
 done
done
