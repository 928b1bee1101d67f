#!/bin/bash
# usage: pmc_run.sh <tag> "<counters>" [bench.py args]  -> gpurun_out/prof/pmc_<tag>.txt  (one stream, counters only: no other trace domain)
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --workers 1 --no-cpu-baseline --no-extra "$@" > /tmp/pmc_$tag.log 2>&1
DB=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python $R/tools/pmc_counters.py $DB $R/gpurun_out/prof/pmc_$tag.txt "python bench.py --steps 2 --warmup 1 --workers 1 --no-cpu-baseline --no-extra $* (one stream)" > /dev/null 2>>/tmp/pmc_$tag.log
tail -2 /tmp/pmc_$tag.log | cut -c1-200
