#!/bin/bash
# usage: OPAL_AB_ARGS="..." ab_run3.sh libA libB ... : like ab_run2.sh with 2 rounds and fewer steps
for i in 1 2; do
  for v in "$@"; do
    OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps ${OPAL_AB_STEPS:-150} --warmup 15 --no-cpu-baseline --no-secondary $OPAL_AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.3f' % d['roofline']['frac'], d['roofline']['rebins_in_timed_region'])"
  done
done
