#!/bin/bash
# Same-box A/B of the policy's check interval for populations in no hurry (2 = prev, 6 = base): quiet host (colliding beams,
# laser-plasma, two-stream), then beside CPU burners (colliding beams and two-stream).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
VARIANTS="prev base" BLOCKS="cb lp ts" scripts/r05_ab_variants.sh cold_quiet > /dev/null 2>&1
pids=""
for i in $(seq 1 ${BURNERS:-24}); do timeout 400 python3 -c "while True: pass" & pids="$pids $!"; done
VARIANTS="prev base" BLOCKS="cb ts" scripts/r05_ab_variants.sh cold_loaded > /dev/null 2>&1
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
echo "== quiet"; cat gpurun_out/r05_ab_cold_quiet/ab.txt; echo "== beside 24 CPU burners"; cat gpurun_out/r05_ab_cold_loaded/ab.txt
