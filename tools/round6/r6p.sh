R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p; mkdir -p $O
cd $R
./tools/ubench/host_sync 16 300 8 100 > $O/host_sync.txt 2>&1
AMD_DIRECT_DISPATCH=0 ./tools/ubench/host_sync 16 300 8 100 > $O/host_sync_dd0.txt 2>&1
cat $O/host_sync.txt; echo dd0; cat $O/host_sync_dd0.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "descri or pair or wxbs or ladder" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run base X=1
run directdisp0 AMD_DIRECT_DISPATCH=0
run base X=1
run directdisp0 AMD_DIRECT_DISPATCH=0
run dd0_hwq8 AMD_DIRECT_DISPATCH=0 GPU_MAX_HW_QUEUES=8
cat $O/env.txt
