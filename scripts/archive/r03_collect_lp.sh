#!/bin/bash
# The laser-plasma bench lines of round 3 (each form in a simulation of its own, warmed up to the same physical time).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r03_bench_laser_plasma_2e7_interaction.json 2>> $O/bench.err
python bench.py --workload laser_plasma --npc 2000 --steps 2000 --no-cpu-baseline > $O/r03_bench_laser_plasma_2e7_approach.json 2>> $O/bench.err
python bench.py --workload laser_plasma --npc 10000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r03_bench_laser_plasma_1e8_interaction.json 2>> $O/bench.err
echo done
