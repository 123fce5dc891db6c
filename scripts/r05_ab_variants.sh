#!/bin/bash
# Same-box A/B of library variants (opal_amd/lib/libvar_<name>.so: scripts/ab_build.sh for a commit, scripts/ab_variant.sh for
# compiler flags) on the driver's N = 1 line: configs[2] contract run + default path, and the colliding-beams block.
# usage (through gpurun, from the repo root): VARIANTS="r04 base qw5" BLOCKS="ts cb" scripts/r05_ab_variants.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_ab_${1:-variants}
mkdir -p $O
for rep in 1 2; do
 for v in ${VARIANTS:-r04 base}; do
  lib=$R/opal_amd/lib/libvar_$v.so; [ $v = base ] && lib=$R/opal_amd/lib/libopal_amd.so
  for b in ${BLOCKS:-ts cb}; do
   if [ $b = ts ]; then
    OPAL_AMD_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-secondary --no-laser-plasma --no-colliding-beams --no-live-traffic --steps 200 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['default_path']; print('$v configs2 contract', round(d['ms_per_step'],4), 'push', round(d['roofline']['avg_launch_ms'],4), ' default path', round(p['ms_per_step'],4), 'push', round(p['roofline']['avg_launch_ms'],4), flush=True)" | tee -a $O/ab.txt
   elif [ $b = cb ]; then
    OPAL_AMD_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-laser-plasma --no-default-path --no-live-traffic --steps 20 --warmup 5 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['colliding_beams']; k=c['kernels']; print('$v colliding beams', round(c['ms_per_step'],4), 'ms/step', '%.3e' % c['value'], ' electron push', round(k['electron_push']['avg_launch_ms'],4), 'frac', round(k['electron_push']['frac'],3), ' photon push', round(k['photon_push']['avg_launch_ms'],4), ' k_radiate', round(k['k_radiate']['avg_launch_ms'],4), 'share', round(c['k_radiate_share_of_gpu_time'],3), ' whole-step GB/s', round(c['roofline']['algorithmic_GBps_over_whole_step']), flush=True)" | tee -a $O/ab.txt
   elif [ $b = lp ]; then
    OPAL_AMD_LIB=$lib python3 $R/bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path --no-colliding-beams --no-live-traffic 2>>$O/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v laser-plasma 2e7', round(d['ms_per_step'],4), 'ms/step', '%.3e' % d['value'], ' push', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), ' whole step', round(d['config']['frac_of_hbm_roofline_whole_step'],3), ' rebins', d['roofline']['rebins_in_timed_region'], flush=True)" | tee -a $O/ab.txt
   fi
  done
 done
done
