#!/bin/bash
# usage: ab_bench.sh <variant> ...   -- bench.py (no CPU baseline, no extras) under each library variant ("base" = the tree's), twice round robin
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L python $R/bench.py --no-cpu-baseline --no-extra 2>/dev/null | python $R/tools/bench_line.py $v
done; done
