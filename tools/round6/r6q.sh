R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6q; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_$lab.txt | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run flag X=1
run runtime MODSX_HOST_WAIT=runtime
run flag X=1
run runtime MODSX_HOST_WAIT=runtime
cat $O/env.txt
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
