R=$GRAFT_REPO_ROOT; cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
for i in 1 2; do timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder; done
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py headline
python tools/latency.py 2>&1 | tail -3 | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_views.py -x -q 2>&1 | grep -E "passed|failed"
