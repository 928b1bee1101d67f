#!/bin/bash
# marginal cost of a kernel class under the bench's 16 streams: bench.py with the class launched twice (library built by
# tools/build_variant.sh dup "-DMODSX_DUP_BUILD"); usage: ab_dup.sh <mask> ...   (bit = KClass, engine.hpp; 0 = nothing doubled)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for m in "$@"; do
  MODSX_DUP=$m MODSX_LIB=$R/mods_amd/libmodsx_dup.so python $R/bench.py --no-cpu-baseline --no-extra 2>/dev/null | python $R/tools/bench_line.py dup=$m
done; done
