#!/bin/bash
# value against the number of timed steps (the edge effects of a run: ramp-up of 16 contexts, the drain at the end)
R=$GRAFT_REPO_ROOT
for k in "$@"; do
  python $R/bench.py --no-cpu-baseline --no-extra --steps $k --warmup 3 2>/dev/null | python $R/tools/bench_line.py steps=$k
done
