R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6m; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_shard.py -x -q -m gpu -k "owner" > $O/pytest_shard.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair or orient or descri" > $O/pytest.txt 2>&1
MODSX_BENCH_NO_UPLOAD_LEG=1 timeout 600 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
grep -n "passed\|failed" $O/pytest_shard.txt $O/pytest.txt
grep -A45 "module chain" $O/host_profile.txt | cut -c1-250
python tools/bench_line.py sampled < $O/host_sampler.log 2>/dev/null | tail -1
