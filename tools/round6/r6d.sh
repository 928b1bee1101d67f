R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
bash tools/prof_cmd.sh one "k_" python $R/bench.py --steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra > $O/prof_one.txt 2>&1
tail -3 $O/pytest.txt; python tools/bench_line.py base < $O/bench_default.json; cat $O/prof_one.txt | cut -c1-200
