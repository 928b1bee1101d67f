R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6t; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run flag_small MODSX_HOST_COPY_MAX=65536
run flag_rtcopy MODSX_HOST_COPY=runtime
run runtime MODSX_HOST_WAIT=runtime
run flag_kcopy X=1
cat $O/env.txt
