#!/bin/bash
# configs[1] counter passes (the bench line's `secondary`), the colliding-beams run (configs[4]: electron push with the
# optical depth, photon push, k_radiate) under the HBM counters, and the linked-step traces.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
cd $R
scripts/profile_bench.sh r03_c1_raw --cells 4096 --npc 2441 > $O/c1.log 2>&1
OPAL_PROFILES_DIR=$O/c1_summary python3 scripts/summarise_profile.py gpurun_out/r03_c1_raw r03_configs1 > $O/c1_summary.log 2>&1
rm -rf gpurun_out/r03_c1_raw
echo "configs[1] done"
cd /tmp
NPC=20000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cb_kt -o kt -- python3 $R/scripts/bench_colliding_beams.py > $O/r03_colliding_beams_1e7_steps.log 2> $O/cb_kt.log
NPC=20000 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cb_f -o f -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/cb_f.log
NPC=20000 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cb_w -o w -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/cb_w.log
python3 $R/scripts/summarise_pmc_generic.py $O/cb_kt $O/cb_f $O/cb_w $O/r03_colliding_beams_1e7_pmc_hbm.json "scripts/bench_colliding_beams.py, NPC=20000 (1e7 electrons -> 4.7e7 photons), the whole 3000-step run." > $O/cb_summary.log
cp $(find $O/cb_kt -name "*kernel_stats.csv") $O/r03_colliding_beams_1e7_kernel_stats.csv
rm -rf $O/cb_kt $O/cb_f $O/cb_w
echo "colliding beams done"
cd $R
scripts/trace_linked.sh r03x/tl > $O/tl.log 2>&1
echo "traces done"
