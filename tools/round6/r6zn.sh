R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6zn; mkdir -p $O
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run base X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run sdma0 HSA_ENABLE_SDMA=0
run hwq8 GPU_MAX_HW_QUEUES=8
run base X=1
cat $O/env.txt
