#!/bin/bash
# Counter passes of BASELINE configs[4] (examples/colliding_beams.yaml's physics, npc 20 000: 1e7 electrons, ~4.7e7 photons)
# on one MI355X, the whole 3000-step run: --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in passes of their own
# (every kernel of the step: electron push with the optical depth, photon push, k_radiate, re-binnings, field solve).
# usage (through gpurun, from the repo root): scripts/profile_cb.sh <tag>     ->  gpurun_out/<tag>_{kernel_stats.csv,pmc_hbm.json,steps.log}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04_colliding_beams_1e7}
O=$R/gpurun_out/cb_raw
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/scripts/bench_colliding_beams.py > $R/gpurun_out/${TAG}_steps.log 2> $O/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/f.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/scripts/bench_colliding_beams.py > /dev/null 2> $O/w.log
python3 $R/scripts/summarise_pmc_generic.py $O/kt $O/f $O/w $R/gpurun_out/${TAG}_pmc_hbm.json "examples/colliding_beams.yaml physics, npc 20000 (1e7 electrons), whole run, scripts/profile_cb.sh."
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $O
