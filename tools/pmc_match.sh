#!/bin/bash
# usage: pmc_match.sh <tag> "<counters>" [bench_match args]  -> gpurun_out/prof/pmc_match_<tag>.txt  (counters only, no other trace domain)
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm_$tag
rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcm_$tag -o p -- python $R/tools/bench_match.py "$@" > /tmp/pmcm_$tag.log 2>&1
DB=$(find /tmp/pmcm_$tag -name "*.db" | head -1)
python $R/tools/pmc_counters.py $DB $R/gpurun_out/prof/pmc_match_$tag.txt "python tools/bench_match.py $*" 2>>/tmp/pmcm_$tag.log | grep "kernel \|k_match"
tail -3 /tmp/pmcm_$tag.log
