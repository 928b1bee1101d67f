R=$GRAFT_REPO_ROOT; cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_views.py -x -q 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do
MODSX_LIB=$R/mods_amd/libmodsx_old.so LOCAL_WORLD_SIZE=8 taskset -c 0,1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py old_2cpu
LOCAL_WORLD_SIZE=8 taskset -c 0,1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py new_2cpu
done
