R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6u; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py $EXTRA --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
export MODSX_HOST_COPY_MAX=65536
run flag_small X=1
run flag_thensync MODSX_WAIT_THEN_SYNC=1
EXTRA="--workers 20" run flag_w20 X=1
run runtime MODSX_HOST_WAIT=runtime
run flag_nap5 MODSX_WAIT_NAP_MAX_US=5
cat $O/env.txt
