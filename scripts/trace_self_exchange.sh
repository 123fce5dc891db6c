#!/bin/bash
# kernel trace of the multi-GPU step rehearsed on one GPU (bench.py --self-exchange): what runs between two pushes
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-trace_self}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --self-exchange --no-cpu-baseline --no-secondary --steps 60 --warmup 10 ${OPAL_TRACE_ARGS} > $O/bench.json 2> $O/kt.log
echo done
