R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6za; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in auto auto; do
  rm -rf /tmp/rp_def
  MODSX_HOST_WAIT=$mode timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_def -o p -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /tmp/rp_def.log 2>&1
  echo "mode $mode rc $?" >> $O/summary.txt
  grep -E "^\{" /tmp/rp_def.log | python $R/tools/bench_line.py $mode >> $O/summary.txt 2>&1
  grep -v "^{" /tmp/rp_def.log | tail -40 > $O/log_$mode.txt
done
cat $O/summary.txt
cd $R; ROUND_TAG=r06 bash tools/regen_default_profile.sh
timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder
timeout 300 python -m pytest tests/test_gpu_views.py -x -q 2>&1 | tail -2
