#!/bin/bash
# Profile passes of a bench.py command on one MI355X (run through gpurun from the repo root):
#   scripts/profile_bench.sh <out-dir-under-gpurun_out> [bench.py arguments...]
# Each rocprofv3 pass is its own run of the SAME command: --kernel-trace --stats alone (full step counts), then one
# --pmc pass per counter group (5 steps).  rocprofv3 is given python3 itself, never a wrapper.
# scripts/summarise_profile.py turns the output directory into the tracked files under profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1
shift
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu-baseline --no-secondary --no-live-traffic $*"
echo "$ARGS" > $O/args.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_under_kt.json 2> $O/kt.log
echo "kernel-trace done"
SHORT="--steps 5 --warmup 2"
case " $* " in *" --steps "*) SHORT="";; esac
echo "$ARGS $SHORT" > $O/args_pmc.txt
rocprofv3 --kernel-include-regex "k_push" --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $ARGS $SHORT > /dev/null 2> $O/pmc_f.log
rocprofv3 --kernel-include-regex "k_push" --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $ARGS $SHORT > /dev/null 2> $O/pmc_w.log
echo "hbm done"
rocprofv3 --kernel-include-regex "k_push" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py $ARGS $SHORT > /dev/null 2> $O/pmc_sq.log
echo "sq done"
rocprofv3 --kernel-include-regex "k_push" --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o sq2 -- python3 $R/bench.py $ARGS $SHORT > /dev/null 2> $O/pmc_sq2.log
echo "sq2 done"
