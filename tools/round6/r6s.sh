R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6s; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run flag_kcopy16 X=1
run flag_kcopy16_small MODSX_HOST_COPY_MAX=65536
run flag_rtcopy MODSX_HOST_COPY=runtime
run runtime MODSX_HOST_WAIT=runtime
run flag_kcopy16 X=1
cat $O/env.txt
timeout 600 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
python tools/bench_line.py sampled < $O/host_sampler.log 2>/dev/null | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
