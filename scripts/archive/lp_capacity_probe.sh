#!/bin/bash
# Does the link's particle capacity cost anything per step?  configs[3] on 2 ranks sharing the GPU, inside the interaction.
cd $GRAFT_REPO_ROOT
for cap in 20000 1280000 20000 1280000; do
OPAL_BENCH_LP_CAPACITY=$cap python bench.py --gpus 2 --workload laser_plasma --npc 10000 --warmup 21000 --steps 2000 --no-default-path --no-cpu-baseline 2>gpurun_out/lpcap.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('capacity $cap: %.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % r['avg_launch_ms'], 'rebins', r['rebins_in_timed_region'], r['other_kernels_avg_ms'], 'launches', r['launches_total'])"
done
