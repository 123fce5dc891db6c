#!/bin/bash
# Why the colliding-beams block's mailboxes do not come up after a laser-plasma block with moving cuts (N ranks on the card)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_cbl
mkdir -p $O
cd $R
python3 bench.py --gpus ${RANKS:-4} --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/run.json 2> $O/run.err
grep -n "bench\]" $O/run.err | cut -c1-600
python3 scripts/r05_show_blocks.py $O/run.json 2>&1 | grep -v "^       {" | cut -c1-300
