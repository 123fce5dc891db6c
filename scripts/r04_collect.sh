#!/bin/bash
# Round-4 bench lines on one MI355X (through gpurun, from the repo root): what DESIGN.md 6 quotes.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
STAGE=${1:-all}
if [ $STAGE = all ] || [ $STAGE = default ]; then
python bench.py > $O/r04_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r04_bench_driver_command.json 2>> $O/bench.err
echo "default done"
fi
if [ $STAGE = all ] || [ $STAGE = share ]; then
for LINK in peer rccl; do
  OPAL_BENCH_LINK=$LINK python bench.py --self-exchange --cells 4096 --npc 1526 --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary > $O/r04_bench_self_exchange_${LINK}_one_eighth.json 2>> $O/bench.err
done
OPAL_BENCH_LINK=peer python bench.py --self-exchange --elision on --cells 4096 --npc 1526 --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary > $O/r04_bench_self_exchange_peer_one_eighth_default_path.json 2>> $O/bench.err
OPAL_BENCH_LINK=peer python bench.py --self-exchange --steps 400 --warmup 40 --no-cpu-baseline --no-secondary > $O/r04_bench_self_exchange_peer_1e8.json 2>> $O/bench.err
echo "self-exchange done"
fi
if [ $STAGE = all ] || [ $STAGE = ranks ]; then
for N in 2 4 6; do
  python bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline > $O/r04_bench_${N}ranks_on_1gpu_rehearsal.json 2>> $O/bench.err
done
echo "rehearsals done"
fi
if [ $STAGE = all ] || [ $STAGE = lp ]; then
python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r04_bench_laser_plasma_2e7_interaction.json 2>> $O/bench.err
python bench.py --workload laser_plasma --npc 10000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r04_bench_laser_plasma_1e8_interaction.json 2>> $O/bench.err
echo "laser-plasma done"
python bench.py --unfused --no-cpu-baseline --no-secondary > $O/r04_bench_unfused_operator_sequence.json 2>> $O/bench.err
fi
echo "all done"
