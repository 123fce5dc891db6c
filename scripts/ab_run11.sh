#!/bin/bash
# usage: ab_run11.sh libA libB ... : laser-plasma interaction only, two rounds
for i in 1 2; do
for v in "$@"; do
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v laser-plasma interaction', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], d['roofline']['rebins_in_timed_region'])"
done
done
