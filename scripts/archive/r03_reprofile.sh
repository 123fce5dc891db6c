#!/bin/bash
# Everything profiles/traffic.json depends on, again (after a change to the kernel sources): the driver's default command,
# configs[1] and the laser-plasma interaction under rocprofv3 (kernel trace + the separate counter passes), and the
# linked-step traces.  Through gpurun from the repo root; summaries that need the raw files of long runs are made on the box.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r03x
mkdir -p $O
scripts/profile_bench.sh r03_prof > $O/prof.log 2>&1
echo "default command done"
scripts/profile_bench.sh r03_c1_raw --cells 4096 --npc 2441 > $O/c1.log 2>&1
OPAL_PROFILES_DIR=$O/c1_summary python3 scripts/summarise_profile.py gpurun_out/r03_c1_raw r03_configs1 > $O/c1_summary.log 2>&1
rm -rf gpurun_out/r03_c1_raw
echo "configs[1] done"
scripts/profile_lp.sh r03_laser_plasma_interaction > $O/lp.log 2>&1
echo "laser-plasma done"
scripts/trace_linked.sh r03x/tl > $O/tl.log 2>&1
echo "traces done"
