#!/bin/bash
# rocprofv3 kernel-trace summary of a bench.py run (through gpurun).  $1 = tag, rest = bench.py args
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpb_$tag
rocprofv3 --kernel-trace --stats -d /tmp/rpb_$tag -o p -- python $R/bench.py "$@" > /tmp/rpb_$tag.log 2>&1
DB=$(find /tmp/rpb_$tag -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/prof/bench_$tag.txt "python bench.py $*" > /dev/null
head -1 /tmp/rpb_$tag.log | cut -c1-160
