#!/bin/bash
# usage: ab_run7.sh libA libB ... : same-box two-stream runs per build at one GPU's share of configs[2] at N = 8 and N = 4
for v in "$@"; do
  for sz in "4096 1526" "8192 1526"; do
    set -- $sz
    OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 600 --warmup 15 --no-cpu-baseline --no-secondary --cells $1 --npc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v cells $1 npc $2', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'])"
  done
done
