#!/bin/bash
# usage: pmc_run.sh <tag> <counter> [<counter> ...]  -> gpurun_out/pmc_<tag>.txt  (one stream, counters only)
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --workers 1 --batch 8 --no-cpu-baseline > /tmp/pmc_$tag.log 2>&1
DB=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python $R/tools/pmc_counters.py $DB $R/gpurun_out/pmc_$tag.txt "python bench.py --steps 2 --warmup 1 --workers 1 --batch 8 --no-cpu-baseline (one stream)" > /dev/null 2>>/tmp/pmc_$tag.log
tail -5 /tmp/pmc_$tag.log > $R/gpurun_out/pmc_$tag.log
