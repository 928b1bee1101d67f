#!/bin/bash
# SQ counters of the one-stream default bench (profiles/rNN_pmc_sq.txt): quick A/B of a kernel change
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-pmcone}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
rm -rf /tmp/rq1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/rq1 -o p -- python $R/bench.py --steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra > /tmp/rq1.log 2>&1
python $R/tools/pmc_counters.py $(find /tmp/rq1 -name "*.db" | head -1) $OUT/pmc_sq.txt "one stream" | cut -c1-230 | head -${2:-10}
