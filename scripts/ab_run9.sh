#!/bin/bash
# usage: ab_run9.sh libA libB ... : per build, the linked step at one GPU's share of configs[2] at N = 8, the laser-plasma
# interaction phase, and the linked step at 1e8
for v in "$@"; do
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --self-exchange --steps 600 --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1.25e7 linked  ', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'], d['roofline']['rebins_in_timed_region'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v laser-plasma interaction', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], d['roofline']['rebins_in_timed_region'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --self-exchange --steps 400 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1e8 linked     ', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'], d['roofline']['rebins_in_timed_region'])"
done
