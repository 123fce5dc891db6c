#!/bin/bash
# Everything profiles/traffic.json depends on, with the round's final kernel sources: the driver's default command, configs[1]
# and the laser-plasma interaction under rocprofv3 (kernel trace + the separate counter passes), the colliding-beams run,
# the linked-step traces.  Through gpurun from the repo root; every summary is made on the box into gpurun_out/r05_summary
# (one directory, so that traffic.json collects all three workloads), which is then copied into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
S=$R/gpurun_out/r05_summary
mkdir -p $S
export OPAL_PROFILES_DIR=$S
STAGE=${1:-all}
if [ $STAGE = all ] || [ $STAGE = default ]; then
scripts/profile_bench.sh r05_prof > $S/prof.log 2>&1
python3 scripts/summarise_profile.py gpurun_out/r05_prof r05 > $S/prof_summary.log 2>&1
rm -rf gpurun_out/r05_prof
echo "default command done"
scripts/profile_bench.sh r05_c1_raw --cells 4096 --npc 2441 > $S/c1.log 2>&1
python3 scripts/summarise_profile.py gpurun_out/r05_c1_raw r05_configs1 > $S/c1_summary.log 2>&1
rm -rf gpurun_out/r05_c1_raw
echo "configs[1] done"
fi
if [ $STAGE = all ] || [ $STAGE = lp ]; then
scripts/profile_bench.sh lp_raw --workload laser_plasma --npc 2000 --warmup 21000 --steps 2000 --no-default-path > $S/lp.log 2>&1
python3 scripts/summarise_profile.py gpurun_out/lp_raw r05_laser_plasma_interaction --last 2000 > $S/lp_summary.log 2>&1
rm -rf gpurun_out/lp_raw
echo "laser-plasma done"
fi
if [ $STAGE = all ] || [ $STAGE = cb ]; then
scripts/profile_cb.sh r05_colliding_beams_1e7 > $S/cb.log 2>&1
mv gpurun_out/r05_colliding_beams_1e7_* $S/
echo "colliding beams done"
fi
if [ $STAGE = all ] || [ $STAGE = trace ]; then
scripts/trace_linked.sh r05_tl > $S/tl.log 2>&1
cat gpurun_out/r05_tl/timeline_peer.txt gpurun_out/r05_tl/timeline_rccl.txt > $S/r05_linked_step_trace_raw.txt 2>/dev/null || true
python3 - <<'PY' >> $S/r05_linked_step_trace_raw.txt
import glob, json, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
print("==== the same runs without the tracer (2000 timed steps each, alternating)")
for kind in ("peer", "rccl"):
    for line in open(os.path.join(R, "gpurun_out", "r05_tl", f"untraced_{kind}.json")):
        if line.strip().startswith("{"):
            d = json.loads(line)
            print(f"  {kind}  {d['ms_per_step']:.4f} ms/step  {d['value']:.3e} particle-updates/s   (push kernel {d['roofline']['avg_launch_ms']:.4f} ms)")
PY
echo "traces done"
fi
