#!/bin/bash
# The laser-plasma block on N ranks sharing the card with the link's and the re-cut's diagnostics (one run per argument: env assignments)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_lp6
mkdir -p $O
cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs OPAL_LINK_DEBUG=1 OPAL_RECUT_DEBUG=1 python3 bench.py --gpus ${RANKS:-6} --steps 20 --warmup 5 --no-cpu-baseline --no-colliding-beams --no-live-traffic > $O/run$i.json 2> $O/run$i.err
  echo "run $i ($envs): $(grep -c abandoned $O/run$i.err) abandoned"
  grep -n "recut\]\|link\]\|abandoned" $O/run$i.err | tail -${TAIL:-40} | cut -c1-400
  [ -n "$STOP_AT_FAILURE" ] && [ "$(grep -c abandoned $O/run$i.err)" != 0 ] && break
done
