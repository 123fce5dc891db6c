#!/bin/bash
# The colliding-beams block (four ~85 us kernels per step) on a quiet host and beside CPU burners that use up the box's CPU quota,
# with the HIP runtime's defaults and with its signal pool / active-wait knobs: why the block read 0.46-0.56 ms per step on some
# boxes where its kernels add up to 0.34.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_cb_host
mkdir -p $O
cd $R
one() {  # label, env...
  label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-laser-plasma --no-default-path --no-live-traffic --steps 20 --warmup 5 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('colliding_beams') or {}; k=c.get('kernels',{})
print('$label', 'two-stream', round(d['ms_per_step'],4), ' colliding beams', round(c.get('ms_per_step',0),4), 'ms/step; kernels add up to', round(sum(v['avg_launch_ms'] for v in k.values()),4), flush=True)" | tee -a $O/ab.txt
}
for load in quiet loaded; do
  pids=""
  if [ $load = loaded ]; then
    for i in $(seq 1 ${BURNERS:-24}); do timeout 300 python3 -c "while True: pass" & pids="$pids $!"; done
  fi
  one "$load defaults" X=1
  one "$load ROC_SIGNAL_POOL_SIZE=1024" ROC_SIGNAL_POOL_SIZE=1024
  one "$load ROC_ACTIVE_WAIT_TIMEOUT=500" ROC_ACTIVE_WAIT_TIMEOUT=500
  one "$load both" ROC_SIGNAL_POOL_SIZE=1024 ROC_ACTIVE_WAIT_TIMEOUT=500
  for p in $pids; do kill $p 2>/dev/null; done
  wait 2>/dev/null
done
