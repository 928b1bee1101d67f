R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_views.py tests/test_gpu_shard.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
export MODSX_BENCH_NO_UPLOAD_LEG=1
for i in 1 2; do
MODSX_MSER_DEVICE_SORT=0 timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_hostsort
timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_devsort
done
