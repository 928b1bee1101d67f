#!/bin/bash
# rocprofv3 kernel-trace summary of the stand-alone matcher benchmark (run through gpurun).  $1 = tag, rest = bench_match args
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$tag
rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag -o p -- python $R/tools/bench_match.py "$@" > /tmp/rp_$tag.log 2>&1
tail -2 /tmp/rp_$tag.log
DB=$(find /tmp/rp_$tag -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/prof/match_$tag.txt "python tools/bench_match.py $*" | grep -i "kernel \|match"
