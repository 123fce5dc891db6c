#!/bin/bash
# End-of-round check on one MI355X: smoke, the whole -m gpu suite, the two default bench lines.
set -e
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r03; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $R/gpurun_out/t13.log 2>&1 || true; tail -3 $R/gpurun_out/t13.log
python bench.py > $O/r03_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r03_bench_driver_command.json 2>> $O/bench.err
echo benches done
