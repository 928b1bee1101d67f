R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6r; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run flag_kcopy X=1
run flag_rtcopy MODSX_HOST_COPY=runtime
run flag_kcopy_small MODSX_HOST_COPY_MAX=65536
run flag_nap20 MODSX_WAIT_NAP_MAX_US=20
run flag_nap20_rtcopy MODSX_WAIT_NAP_MAX_US=20 MODSX_HOST_COPY=runtime
run flag_spin200 MODSX_WAIT_SPIN_US=200 MODSX_HOST_COPY=runtime
run runtime MODSX_HOST_WAIT=runtime
cat $O/env.txt
