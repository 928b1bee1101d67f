#!/bin/bash
# usage: match_env.sh "<ENV=..>" ...   -- rocprof kernel averages of tools/bench_match.py (31 views) under each environment setting
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); rm -rf /tmp/rpe_$i
  env $e rocprofv3 --kernel-trace --stats -d /tmp/rpe_$i -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 10 > /tmp/rpe_$i.log 2>&1
  DB=$(find /tmp/rpe_$i -name "*.db" | head -1)
  echo "== $e"; python $R/tools/rocprof_summary.py $DB /tmp/se_$i.txt "x" | grep -i "k_match"
done
