#!/bin/bash
# scripts/r05_signal_pool_probe.py beside CPU burners: runtime default, ROC_SIGNAL_POOL_SIZE from outside, and set by the process
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
pids=""
for i in $(seq 1 24); do timeout 120 python3 -c "while True: pass" & pids="$pids $!"; done
for rep in 1 2; do
python3 scripts/r05_signal_pool_probe.py 2>/dev/null | tail -1
ROC_SIGNAL_POOL_SIZE=1024 python3 scripts/r05_signal_pool_probe.py 2>/dev/null | tail -1
python3 scripts/r05_signal_pool_probe.py inproc 2>/dev/null | tail -1
done
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
