#!/bin/bash
# Same-box A/B of two builds (scripts/ab_build.sh: libvar_base = HEAD; libopal_amd = the working tree) at the sizes that
# matter: one GPU's share of configs[2] at N = 8 through the peer self-exchange, configs[1], configs[2], the laser-plasma
# interaction.  Through gpurun from the repo root: bash scripts/ab_sizes.sh
cd $GRAFT_REPO_ROOT
run() { OPAL_AMD_LIB=$PWD/opal_amd/lib/$1.so python bench.py --steps ${STEPS:-150} --warmup 15 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; dp=d.get('default_path')
print('$1', '%.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % r['avg_launch_ms'], ('| default %.4e %.4f ms kernel %.4f' % (dp['value'], dp['ms_per_step'], dp['roofline']['avg_launch_ms'])) if dp else '')"; }
echo "== N=8 share, peer self-exchange"; for i in 1 2; do for v in libvar_base libopal_amd; do STEPS=600 run $v "--self-exchange --cells 4096 --npc 1526"; done; done
echo "== configs[1]"; for i in 1 2; do for v in libvar_base libopal_amd; do run $v "--cells 4096 --npc 2441"; done; done
echo "== configs[2] 1e8"; for i in 1; do for v in libvar_base libopal_amd; do run $v ""; done; done
echo "== LP interaction"; for i in 1; do for v in libvar_base libopal_amd; do run $v "--workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-default-path"; done; done
