#!/bin/bash
# Round-5 bench lines on one MI355X (through gpurun, from the repo root): what DESIGN.md 6 quotes.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
STAGE=${1:-all}
if [ $STAGE = all ] || [ $STAGE = default ]; then
python bench.py > $O/r05_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_command.json 2>> $O/bench.err
echo "default done"
fi
if [ $STAGE = all ] || [ $STAGE = share ]; then
OPAL_BENCH_LINK=peer python bench.py --self-exchange --cells 4096 --npc 1526 --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary > $O/r05_bench_self_exchange_peer_one_eighth.json 2>> $O/bench.err
OPAL_BENCH_LINK=peer python bench.py --self-exchange --elision on --cells 4096 --npc 1526 --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary > $O/r05_bench_self_exchange_peer_one_eighth_default_path.json 2>> $O/bench.err
echo "self-exchange done"
fi
for N in 2 4 6; do
if [ $STAGE = all ] || [ $STAGE = ranks ] || [ $STAGE = ranks$N ]; then
  python bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline > $O/r05_bench_${N}ranks_on_1gpu_rehearsal.json 2>> $O/bench.err
  echo "rehearsal $N done"
fi
# the same with the reference's partition, made once at t = 0: what the moving cuts are compared with
if [ $STAGE = static$N ]; then
  OPAL_BENCH_LP_RECUT=0 OPAL_BENCH_CB_RECUT=0 python bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline > $O/r05_bench_${N}ranks_on_1gpu_rehearsal_static_cuts.json 2>> $O/bench.err
  echo "rehearsal $N (static cuts) done"
fi
# both partitions in one line (what a node with a card per rank does by itself): the block is the faster one, the other is kept beside it
if [ $STAGE = both$N ]; then
  OPAL_BENCH_BOTH_PARTITIONS=1 python bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline > $O/r05_bench_${N}ranks_on_1gpu_rehearsal_both_partitions.json 2>> $O/bench.err
  echo "rehearsal $N (both partitions) done"
fi
done
if [ $STAGE = all ] || [ $STAGE = lp ]; then
python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r05_bench_laser_plasma_2e7_interaction.json 2>> $O/bench.err
python bench.py --workload laser_plasma --npc 10000 --warmup 21000 --steps 2000 --no-cpu-baseline > $O/r05_bench_laser_plasma_1e8_interaction.json 2>> $O/bench.err
echo "laser-plasma done"
fi
echo "all done"
