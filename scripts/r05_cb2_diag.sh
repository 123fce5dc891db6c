#!/bin/bash
# The colliding-beams block on 2 ranks sharing the card, static cuts and moving cuts, with the re-cut's and the link's diagnostics.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_cb2
mkdir -p $O
cd $R
[ -n "$SKIP_STATIC" ] || OPAL_BENCH_CB_RECUT=0 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-laser-plasma --no-live-traffic > $O/static.json 2> $O/static.err
echo "static done: $(grep -c 'abandoned' $O/static.err) abandoned"
OPAL_BENCH_CB_VERBOSE=1 OPAL_RECUT_DEBUG=1 OPAL_LINK_DEBUG=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-laser-plasma --no-live-traffic > $O/moving.json 2> $O/moving.err
echo "moving done: $(grep -c 'abandoned' $O/moving.err) abandoned"
grep -n "recut\]\|link\]\|abandoned\|warm-up\|segment" $O/moving.err | tail -60 | cut -c1-260
