R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6zg; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_views.py -x -q -k "flag_word" 2>&1 | grep -E "passed|failed|Error" | tail -3
export MODSX_BENCH_NO_UPLOAD_LEG=1
MODSX_HOST_WAIT=alternate timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>$O/alt.err | python tools/bench_line.py alternate
grep -o '"parity": {[^}]*}' $O/alt.err | head -1
MODSX_HOST_WAIT=alternate timeout 600 python bench.py --config ladder --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_alternate
(MODSX_HOST_WAIT=alternate timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_alt.log 2>&1; echo "pytest(alternate) rc $?" >> $O/pytest_alt.log); grep -E "passed|failed|rc " $O/pytest_alt.log | tail -3
