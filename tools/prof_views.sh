#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profv
rocprofv3 --kernel-trace --stats -d /tmp/profv -o p -- python $R/bench.py --tilts 1,2,3,4,6 --steps 3 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline > /tmp/profv.log 2>&1
DB=$(find /tmp/profv -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/prof_views.txt "python bench.py --tilts 1,2,3,4,6 --steps 3 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline (one stream, 8 views per image)" > /dev/null
tail -1 /tmp/profv.log | cut -c1-200
