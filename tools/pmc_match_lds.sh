#!/bin/bash
# LDS-side counters of the stand-alone matcher at 24 k x 24 k (separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$i -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3 > /tmp/pl_$i.log 2>&1
  DB=$(find /tmp/pl_$i -name "*.db" | head -1)
  if [ -z "$DB" ]; then echo "set $i: no database ($(tail -2 /tmp/pl_$i.log | tr '\n' ' '))"; continue; fi
  python $R/tools/pmc_counters.py $DB /tmp/pl_$i.txt "x" > /dev/null 2>&1
  grep -E "^kernel|k_match" /tmp/pl_$i.txt | cut -c1-230
done
