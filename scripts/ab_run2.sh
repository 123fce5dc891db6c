#!/bin/bash
# usage: ab_run2.sh libA libB : alternates bench runs of two builds on the same GPU box
for i in 1 2 3; do
  for v in "$@"; do
    OPAL_AMD_LIB=$PWD/opal_amd/lib/$v.so python bench.py --steps 150 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.4e' % d['value'], '%.4f' % d['roofline']['avg_launch_ms'])"
  done
done
