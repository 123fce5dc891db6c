#!/bin/bash
# usage: ab_run2.sh libA libB ... : alternates bench runs of several builds (scripts/ab_variant.sh) on the same GPU box
# (OPAL_AB_ARGS: extra bench.py arguments, e.g. "--cells 4096 --npc 2441" for configs[1]; OPAL_AB_REPS: rounds, default 3)
for i in $(seq 1 ${OPAL_AB_REPS:-3}); do
  for v in "$@"; do
    OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-secondary $OPAL_AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; dp=d.get('default_path')
print('$v', '%.4e' % d['value'], 'kernel %.4f ms' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'rebins %d' % r['rebins_in_timed_region'], ('| default %.4e' % dp['value']) if dp else '')"
  done
done
