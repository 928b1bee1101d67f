#!/bin/bash
# the two shapes of k_match_sweep1 (MODSX_SWEEP1_FAT=0 / 1 forces one; unset = the size rule) at three sizes, alternating
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 1 auto; do
  for spec in "m10k:" "m24k:--tilts 1,2,4,6,8 --phi 120" "m48k:--tilts 1,2,4,6,8 --phi 120 --rep 2"; do
    tag=${spec%%:*}; a=${spec#*:}
    rm -rf /tmp/rq_$tag
    if [ $v = auto ]; then unset MODSX_SWEEP1_FAT; else export MODSX_SWEEP1_FAT=$v; fi
    rocprofv3 --kernel-trace --stats -d /tmp/rq_$tag -o p -- python $R/tools/bench_match.py $a --reps 10 > /tmp/rq_$tag.log 2>&1
    DB=$(find /tmp/rq_$tag -name "*.db" | head -1)
    echo "fat=$v $tag $(python $R/tools/rocprof_summary.py $DB /tmp/sq_$tag.txt x | grep k_match | awk '{printf "%s %s  ", $1, $4}')"
  done
done; done
