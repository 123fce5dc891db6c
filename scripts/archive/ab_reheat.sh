#!/bin/bash
# Does a last untimed burst before the timed region (bench.py: REHEAT_STEPS) change what a 20-step timed region measures?
cd $GRAFT_REPO_ROOT
for i in 1 2; do for r in 0 40; do
OPAL_BENCH_REHEAT=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; dp=d['default_path']
print('reheat $r: %.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % r['avg_launch_ms'], '| default %.4e %.4f ms kernel %.4f' % (dp['value'], dp['ms_per_step'], dp['roofline']['avg_launch_ms']))"
done; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; dp=d['default_path']
print('200 steps: %.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % r['avg_launch_ms'], '| default %.4e %.4f ms kernel %.4f' % (dp['value'], dp['ms_per_step'], dp['roofline']['avg_launch_ms']))"
