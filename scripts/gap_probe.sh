set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for V in base chk4; do
  if [ $V = chk4 ]; then export OPAL_AMD_LIB=$R/opal_amd/lib/libvar_chk4.so; fi
  OPAL_BENCH_LINK=peer rocprofv3 --kernel-trace --output-format csv -d $O/$V -o kt -- python3 $R/bench.py --self-exchange --elision on --cells 4096 --npc 1526 --no-cpu-baseline --no-secondary --steps 300 --warmup 20 > $O/b_$V.json 2> $O/$V.log
  python3 $R/scripts/step_timeline.py $O/$V 9 20 > $O/tl_$V.txt; rm -rf $O/$V
done
grep -E "step of|tail" $O/tl_base.txt | head -30; echo ----; grep -E "step of|tail" $O/tl_chk4.txt | head -30
