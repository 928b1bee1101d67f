R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6x; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/err_auto.txt | python tools/bench_line.py "headline_auto" 
timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra > $O/ladder.json 2> $O/ladder.err
python tools/bench_line.py ladder_auto < $O/ladder.json
grep -o '"host_wait": "[^"]*"' $O/ladder.json
MODSX_HOST_WAIT=flag timeout 600 python tools/host_sampler.py $O/host_profile_ladder.txt bench.py --config ladder --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
python tools/bench_line.py ladder_sampled_flag < $O/host_sampler.log
