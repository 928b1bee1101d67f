#!/bin/bash
# step 1 of regen_profiles.sh alone: the kernel-trace summary of the default bench (16 streams)
R=$GRAFT_REPO_ROOT; TAG=${ROUND_TAG:-r05}; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_def
rocprofv3 --kernel-trace --stats -d /tmp/rp_def -o p -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /tmp/rp_def.log 2>&1
DB=$(find /tmp/rp_def -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats_default.txt "python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2  (default workload: 31 views per image, 16 workers x 16 pairs/step; kernels of the 16 streams overlap, durations include time-slicing)" > /dev/null || tail -5 /tmp/rp_def.log
wc -l $OUT/${TAG}_kernel_stats_default.txt
