R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6v; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py $EXTRA --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
export MODSX_HOST_COPY_MAX=65536
run predict X=1
run nopredict MODSX_WAIT_NO_PREDICT=1
run runtime MODSX_HOST_WAIT=runtime
run predict X=1
run runtime MODSX_HOST_WAIT=runtime
cat $O/env.txt
