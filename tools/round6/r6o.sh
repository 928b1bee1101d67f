# host profile of the headline run + the wait-related runtime switches, one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6o; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
timeout 600 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
python tools/bench_line.py sampled < $O/host_sampler.log 2>/dev/null | tail -1
run() { # label, env...
  lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py "$lab" >> $O/env.txt 2>&1 || echo "$lab FAILED" >> $O/env.txt
}
run base X=1
run activewait0 ROC_ACTIVE_WAIT_TIMEOUT=0
run hsaint0 HSA_ENABLE_INTERRUPT=0
run directdisp0 AMD_DIRECT_DISPATCH=0
run cpuwait0 ROC_CPU_WAIT_FOR_SIGNAL=0
cat $O/env.txt
