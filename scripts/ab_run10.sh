#!/bin/bash
# usage: ab_run10.sh libA libB ... : laser-plasma interaction and the default two-stream, alternating builds
for i in 1 2; do
for v in "$@"; do
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v laser-plasma interaction', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], d['roofline']['rebins_in_timed_region'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 300 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1e8 in place   ', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'])"
done
done
