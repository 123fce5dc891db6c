#!/bin/bash
# usage: ab_run4.sh libA libB ... : one same-box round of (a) the default two-stream bench and (b) the laser-plasma
# interaction phase (2e7 particles, steps 21000..23000) per build
for v in "$@"; do
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 300 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v two-stream', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], d['roofline']['rebins_in_timed_region'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v laser-plasma interaction', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], d['roofline']['rebins_in_timed_region'], d['roofline']['other_kernels_avg_ms'])"
done
