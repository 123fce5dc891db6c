#!/bin/bash
# Same-box A/B of library variants (opal_amd/lib/libvar_<name>.so, built from a patched copy of the sources) on the
# laser-plasma interaction: who is re-binned how often (r04_lp_rebin_probe.py, 400 steps) and the bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_variants
mkdir -p $O
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],5), 'ms/step  push', round(r['avg_launch_ms'],5), 'rebins', r['rebins_in_timed_region'], 'x', round(r['other_kernels_avg_ms']['rebin'],4), 'whole-step frac %.4f' % (d['value'] * 80 / 8e12), flush=True)" | tee -a $O/ab.txt; }
for rep in 1 2; do
 for v in ${VARIANTS:-base cc8 twoend twoend_cc8}; do
  lib=$R/opal_amd/lib/libvar_$v.so; [ $v = base ] && lib=$R/opal_amd/lib/libopal_amd.so
  if [ $rep = 1 ]; then
   echo "== $v" | tee -a $O/ab.txt
   OPAL_AMD_LIB=$lib python3 $R/scripts/r04_lp_rebin_probe.py 2000 400 2>>$O/err.log | grep "re-binnings" | tee -a $O/ab.txt
  fi
  OPAL_AMD_LIB=$lib python3 $R/bench.py --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path --no-live-traffic 2>>$O/err.log | line "$v"
 done
done
