#!/bin/bash
# Counter passes INSIDE the laser-plasma interaction (BASELINE configs[3]'s physics on one GPU, 21 000 untimed steps to
# bring the pulse onto the slab, 2000 timed): scripts/profile_bench.sh on that command, summarised on the GPU box (the
# last 2000 launches of each kernel) because the raw files of 23 000 steps are too big to bring back.
# usage (through gpurun, from the repo root): scripts/profile_lp.sh <tag> [npc]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03_laser_plasma_interaction}
NPC=${2:-2000}
$R/scripts/profile_bench.sh lp_raw --workload laser_plasma --npc $NPC --warmup 21000 --steps 2000 --no-default-path
OPAL_PROFILES_DIR=$R/gpurun_out/${TAG}_summary python3 $R/scripts/summarise_profile.py $R/gpurun_out/lp_raw $TAG --last 2000
rm -rf $R/gpurun_out/lp_raw
