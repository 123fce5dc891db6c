#!/bin/bash
# PMC passes of the laser-plasma workload's quiet phase (thermal electron-carbon slab, transverse currents live)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r02_prof_lp}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--workload laser_plasma --npc 2000 --no-cpu-baseline --steps 20 --warmup 5"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-include-regex k_push_pair --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py $ARGS > /dev/null 2> $O/pmc_sq.log
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex k_push_pair --output-format csv -d $O/pmc_sq2 -o sq2 -- python3 $R/bench.py $ARGS > /dev/null 2> $O/pmc_sq2.log
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_push_pair --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $ARGS > /dev/null 2> $O/pmc_f.log
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_push_pair --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $ARGS > /dev/null 2> $O/pmc_w.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_under_kt.json 2> $O/kt.log
echo done
