#!/bin/bash
# Same-box A/B of the re-binning's scatter: straight to the output (OPAL_SORT_STAGED=0) against staged through LDS (1), on a
# population disordered as the laser-plasma interaction's electrons are (scripts/r04_sort_probe.py), then per kernel.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab_sort
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in ${REPS:-1 2}; do
 for m in 0 1; do
  for drift in 13 2; do
   echo -n "staged=$m drift=$drift " | tee -a $O/ab.txt
   OPAL_SORT_STAGED=$m python3 $R/scripts/r04_sort_probe.py 6 $drift 2>>$O/err.log | tee -a $O/ab.txt
  done
 done
done
for m in 0 1; do
 OPAL_SORT_STAGED=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$m -o p -- python3 $R/scripts/r04_sort_probe.py 6 13 >> $O/prof.log 2>&1
 f=$(find $O/prof$m -name '*kernel_stats.csv' | head -1)
 echo "== staged=$m" >> $O/ab.txt
 [ -n "$f" ] && cut -d, -f1-4 $f | grep -i "sort\|fill" | sed 's/(anonymous namespace):://g' | cut -c1-160 >> $O/ab.txt
done
tail -12 $O/ab.txt
