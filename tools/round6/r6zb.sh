R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zb; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
for i in 1 2 3; do
  AMD_DIRECT_DISPATCH=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $O/dd0_$i.json 2> $O/dd0_$i.err
  echo "rc $?" >> $O/summary.txt
  python tools/bench_line.py dd0_$i < $O/dd0_$i.json >> $O/summary.txt 2>&1
  tail -5 $O/dd0_$i.err | cut -c1-300 >> $O/summary.txt
done
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py base >> $O/summary.txt
cat $O/summary.txt
