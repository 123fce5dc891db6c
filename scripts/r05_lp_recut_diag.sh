#!/bin/bash
# Diagnostic: the 4-rank laser-plasma rehearsal (2e7 particles) with the cuts static / moving, and with parts of the re-cut switched.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_lp_diag
mkdir -p $O
N=${RANKS:-4}
run() {  # name, recut spec, extra env...
  name=$1; spec=$2; shift 2
  env "$@" OPAL_BENCH_LP_RECUT=$spec python3 $R/bench.py --gpus $N --workload laser_plasma --npc ${NPC:-2000} --warmup 21000 --steps 2000 --no-cpu-baseline --no-default-path --no-colliding-beams 2>$O/$name.err > $O/$name.json
  python3 - "$name" $O/$name.json <<'PY' | tee -a $O/summary.txt
import json, sys
name, path = sys.argv[1], sys.argv[2]
lines = [l for l in open(path) if l.startswith("{")]
if not lines:
    print(name, "no result")
    sys.exit(0)
d = json.loads(lines[-1])
pr = d.get("per_rank") or d["config"].get("per_rank") or []
print(f"{name}: {d['ms_per_step']:.4f} ms/step {d['value']:.3e}  imbalance {d.get('imbalance_max_over_mean')}  recut {d.get('recut')}"
      f"  push {[round(p['push_avg_ms'], 3) for p in pr]} rebins {[p['rebins'] for p in pr]} particles {[p['particles'] for p in pr]}"
      f" wait_us {[round(p.get('tail_wait_us_per_step', 0)) for p in pr]} room {[p.get('link_room_left_right') for p in pr]}")
PY
}
for what in "$@"; do
  case $what in
    static) run static 0 X=1 ;;
    moving) run moving 50,1.08,32 X=1 ;;
    at_once) run at_once 50,1.08,32 OPAL_RECUT_AT_ONCE=1 ;;
    no_hold) run no_hold 50,1.08,32 OPAL_RECUT_NO_HOLD=1 ;;
    *) run "$what" "$what" X=1 ;;
  esac
done
