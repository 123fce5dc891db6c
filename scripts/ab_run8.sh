#!/bin/bash
# usage: ab_run8.sh libA libB ... : per build, one GPU's share of configs[2] at N = 8 (1.25e7 particles): in place and as
# the linked step (self-exchange), then the 1e8 default
for v in "$@"; do
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 600 --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1.25e7 in place', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --self-exchange --steps 600 --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1.25e7 linked  ', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'], d['roofline']['rebins_in_timed_region'])"
  OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 300 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v 1e8 in place   ', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.4f' % d['ms_per_step'])"
done
