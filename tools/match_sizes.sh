#!/bin/bash
# rocprof kernel averages of the stand-alone matcher at the four sizes of profiles/rNN_match_*.txt (quick look; regen_profiles.sh writes the files)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "m10k:" "m20k:--rep 2" "m24k:--tilts 1,2,4,6,8 --phi 120" "m48k:--tilts 1,2,4,6,8 --phi 120 --rep 2"; do
  tag=${spec%%:*}; a=${spec#*:}
  rm -rf /tmp/rq_$tag
  rocprofv3 --kernel-trace --stats -d /tmp/rq_$tag -o p -- python $R/tools/bench_match.py $a --reps 10 > /tmp/rq_$tag.log 2>&1
  DB=$(find /tmp/rq_$tag -name "*.db" | head -1)
  echo "== $tag $(grep -o 'N=[0-9]* M=[0-9]*' /tmp/rq_$tag.log | tail -1)"; python $R/tools/rocprof_summary.py $DB /tmp/sq_$tag.txt "x" | grep "k_match" | awk '{print $1, $4}'
done
