R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zc; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1 AMD_DIRECT_DISPATCH=0
run() { lab=$1; shift
  for i in 1 2 3 4; do
    env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $O/$lab_$i.json 2> $O/$lab_$i.err
    rc=$?
    if [ $rc -eq 0 ]; then python tools/bench_line.py $lab < $O/$lab_$i.json >> $O/summary.txt 2>&1; else echo "$lab rc $rc: $(grep -h "RuntimeError\|Error" $O/$lab_$i.err | tail -1 | cut -c1-160)" >> $O/summary.txt; fi
  done
}
run flag_allk MODSX_HOST_WAIT=flag MODSX_HOST_COPY_MAX=1000000000
run flag_64k MODSX_HOST_WAIT=flag
run runtime MODSX_HOST_WAIT=runtime
cat $O/summary.txt
