#!/bin/bash
# kernel-trace profile of the single-stream pass; writes gpurun_out/prof_single.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1
rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o p -- python $R/bench.py --steps 6 --warmup 2 --workers 1 --batch 8 --no-cpu-baseline > /tmp/prof1.log 2>&1
DB=$(find /tmp/prof1 -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/prof_single.txt "python bench.py --steps 6 --warmup 2 --workers 1 --batch 8 --no-cpu-baseline  (one stream, 4 pairs per launch set)" > /dev/null
python - <<PY > $R/gpurun_out/prof_blurhess.txt 2>&1
import sqlite3
db = sqlite3.connect("$DB")
for row in db.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels where name like '%blur_hess%' or name like '%k_hessian%' or name like '%resize%' group by name, grid_x, grid_y, grid_z order by name, grid_x desc"):
    print(row)
PY
