#!/bin/bash
# Same-box A/B of chunk lengths (scripts/ab_variant.sh) at one GPU's share of configs[2] at N = 8 and at configs[1].
cd $GRAFT_REPO_ROOT
run() { OPAL_AMD_LIB=$PWD/opal_amd/lib/$1.so python bench.py --steps ${STEPS:-150} --warmup 15 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; dp=d.get('default_path')
print('$1', '%.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % r['avg_launch_ms'], ('| default %.4e %.4f ms kernel %.4f' % (dp['value'], dp['ms_per_step'], dp['roofline']['avg_launch_ms'])) if dp else '')"; }
echo "== N=8 share, peer self-exchange"; for i in 1 2; do for v in libopal_amd libvar_ppt6 libvar_ppt4; do STEPS=1000 run $v "--self-exchange --cells 4096 --npc 1526"; done; done
echo "== configs[1]"; for i in 1 2; do for v in libopal_amd libvar_ppt6 libvar_ppt4; do run $v "--cells 4096 --npc 2441"; done; done
echo "== configs[2]"; for v in libopal_amd libvar_ppt6 libvar_ppt4; do run $v ""; done
