R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6zp; mkdir -p $O
export MODSX_BENCH_NO_UPLOAD_LEG=1
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run base X=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run skipkarg ROC_SKIP_KERNEL_ARG_COPY=1
run fgskarg0 ROC_USE_FGS_KERNARG=0
run aql4k ROC_AQL_QUEUE_SIZE=4096
run sigpool ROC_SIGNAL_POOL_SIZE=256
run base X=1
cat $O/env.txt
