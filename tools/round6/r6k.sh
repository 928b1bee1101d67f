R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
MODSX_BENCH_NO_UPLOAD_LEG=1 timeout 600 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hiptr
MODSX_BENCH_NO_UPLOAD_LEG=1 timeout 600 rocprofv3 --hip-trace --stats -d /tmp/hiptr -o t --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/hiptrace.log 2>&1
find /tmp/hiptr -name "*stats*" | head; for f in $(find /tmp/hiptr -name "*hip_api_stats*.csv" -o -name "*hip_stats*.csv" | head -2); do cp $f $O/; head -40 $f; done
tail -3 $O/hiptrace.log | cut -c1-300
head -150 $O/host_profile.txt | cut -c1-220
