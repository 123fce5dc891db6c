#!/bin/bash
# Round-2 profile passes of the default bench command on one MI355X (run through gpurun from the repo root).
# Each rocprofv3 pass is its own run: --kernel-trace --stats alone, then one --pmc pass per counter group.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r02_prof}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu-baseline --no-secondary $OPAL_PROFILE_ARGS"  # e.g. OPAL_PROFILE_ARGS=--no-elision
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_under_kt.json 2> $O/kt.log
echo "kernel-trace done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py $ARGS --steps 5 --warmup 2 > /dev/null 2> $O/pmc_sq.log
echo "sq done"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o sq2 -- python3 $R/bench.py $ARGS --steps 5 --warmup 2 > /dev/null 2> $O/pmc_sq2.log
echo "sq2 done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $ARGS --steps 5 --warmup 2 > /dev/null 2> $O/pmc_f.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $ARGS --steps 5 --warmup 2 > /dev/null 2> $O/pmc_w.log
echo "hbm done"
