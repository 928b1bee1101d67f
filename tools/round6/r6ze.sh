R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ze; mkdir -p $O
cd $R
export MODSX_HOST_WAIT=flag
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_flag.log 2>&1; echo "pytest(flag) rc $?" >> $O/pytest_flag.log); grep -E "passed|failed|rc " $O/pytest_flag.log | tail -3
python bench.py --no-cpu-baseline --no-extra --shard views 2>/dev/null | python tools/bench_line.py rccl_world1_flag
python bench.py --no-cpu-baseline --no-extra --loopback 8 2>/dev/null | python tools/bench_line.py loopback8_flag
python bench.py --no-cpu-baseline --no-extra --loopback 8 --exchange owner 2>/dev/null | python tools/bench_line.py loopback8_owner_flag
python bench.py --config wxbs --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py wxbs_flag
