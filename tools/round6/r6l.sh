R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
bash tools/ab_bench.sh r6j0 base > $O/ab.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt; cat $O/ab.txt
