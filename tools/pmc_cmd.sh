#!/bin/bash
# usage: pmc_cmd.sh <tag> "<counters>" <kernel filter> <command...>  -> gpurun_out/prof/pmcc_<tag>.txt  (counters only: no other trace domain)
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; filt=$3; shift; shift; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcc_$tag
rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcc_$tag -o p -- "$@" > /tmp/pmcc_$tag.log 2>&1
DB=$(find /tmp/pmcc_$tag -name "*.db" | head -1)
python $R/tools/pmc_counters.py $DB $R/gpurun_out/prof/pmcc_$tag.txt "$*" > /dev/null 2>>/tmp/pmcc_$tag.log
head -3 $R/gpurun_out/prof/pmcc_$tag.txt | tail -1 | cut -c1-220
grep -h "$filt" $R/gpurun_out/prof/pmcc_$tag.txt | cut -c1-220
