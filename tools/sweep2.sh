#!/bin/bash
run() { python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"],1))'; }
for rep in 1 2 3; do
  unset BENCH_NOPROF; echo "prof: $(run)"
  export BENCH_NOPROF=1; echo "noprof: $(run)"
done
