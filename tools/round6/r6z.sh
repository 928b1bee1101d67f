R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in flag runtime auto; do
  rm -rf /tmp/rp_def
  MODSX_HOST_WAIT=$mode timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_def -o p -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /tmp/rp_def.log 2>&1
  echo "mode $mode rc $?" >> $O/summary.txt
  grep -E "^\{" /tmp/rp_def.log | python $R/tools/bench_line.py $mode >> $O/summary.txt 2>&1
  grep -v "^{" /tmp/rp_def.log | tail -40 > $O/log_$mode.txt
  find /tmp/rp_def -name "*.db" | head -1 >> $O/summary.txt
done
cat $O/summary.txt
