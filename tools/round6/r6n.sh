R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6n; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run base X=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run directdisp0 AMD_DIRECT_DISPATCH=0
run activewait50 ROC_ACTIVE_WAIT_TIMEOUT=50
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run cpuwait0 ROC_CPU_WAIT_FOR_SIGNAL=0
run blocksync0 DEBUG_HIP_BLOCK_SYNC=0
run batchsync DEBUG_CLR_BATCH_CPU_SYNC_SIZE=64
run hsaint0 HSA_ENABLE_INTERRUPT=0
run base2 X=1
cat $O/env.txt
