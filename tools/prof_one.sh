#!/bin/bash
# one-stream kernel-trace summary of the default bench (the durations profiles/rNN_kernel_stats_single_stream.txt holds): quick A/B of a kernel change
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-profone}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
rm -rf /tmp/rp1
rocprofv3 --kernel-trace --stats -d /tmp/rp1 -o p -- python $R/bench.py --steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra > /tmp/rp1.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/rp1 -name "*.db" | head -1) $OUT/kernels.txt "one stream" | head -${2:-14}
