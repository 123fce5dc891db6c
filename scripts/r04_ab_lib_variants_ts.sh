#!/bin/bash
# Same-box A/B of library variants on the two-stream workloads: configs[2] (contract run + default path) and one GPU's
# share of it at N = 8 (default path, linked to itself).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_variants_ts
mkdir -p $O
for rep in 1 2; do
 for v in ${VARIANTS:-base twoend}; do
  lib=$R/opal_amd/lib/libvar_$v.so; [ $v = base ] && lib=$R/opal_amd/lib/libopal_amd.so
  OPAL_AMD_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --no-live-traffic --steps 200 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['default_path']; print('$v configs2 contract', round(d['ms_per_step'],4), 'push', round(d['roofline']['avg_launch_ms'],4), ' default path', round(p['ms_per_step'],4), 'push', round(p['roofline']['avg_launch_ms'],4), flush=True)" | tee -a $O/ab.txt
  OPAL_AMD_LIB=$lib python3 $R/bench.py --elision on --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --no-live-traffic --warmup 50 --self-exchange --cells 4096 --npc 1526 --steps 2000 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v share8 default path', round(d['ms_per_step'],5), 'push', round(d['roofline']['avg_launch_ms'],5), flush=True)" | tee -a $O/ab.txt
 done
done
