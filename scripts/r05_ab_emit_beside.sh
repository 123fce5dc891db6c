#!/bin/bash
# Same-box A/B: the emission kernel beside the photons' push (round 5) against behind it (OPAL_EMIT_BESIDE=0), colliding-beams block.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_ab_emit_beside
mkdir -p $O
for rep in 1 2 3; do
 for v in beside behind; do
  B=1; [ $v = behind ] && B=0
  OPAL_EMIT_BESIDE=$B python3 $R/bench.py --no-cpu-baseline --no-laser-plasma --no-default-path --no-live-traffic --steps 20 --warmup 5 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['colliding_beams']; k=c['kernels']; print('$v colliding beams', round(c['ms_per_step'],4), 'ms/step', '%.3e' % c['value'], ' electron push', round(k['electron_push']['avg_launch_ms'],4), 'frac', round(k['electron_push']['frac'],3), ' photon push', round(k['photon_push']['avg_launch_ms'],4), ' k_radiate', round(k['k_radiate']['avg_launch_ms'],4), ' whole-step GB/s', round(c['roofline']['algorithmic_GBps_over_whole_step']), ' photons at the end', c['photons_end'], flush=True)" | tee -a $O/ab.txt
 done
done
