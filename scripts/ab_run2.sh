#!/bin/bash
# usage: ab_run2.sh libA libB ... : alternates bench runs of several builds on the same GPU box
# (OPAL_AB_ARGS: extra bench.py arguments, e.g. "--weak" for configs[1])
for i in 1 2 3; do
  for v in "$@"; do
    OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-secondary $OPAL_AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'], '%.3f' % d['roofline']['frac'])"
  done
done
